"""ctypes binding of librxhip.so (include/rxhip.h).  The library is the product; this module is
plumbing.  It fails loudly when the HIP extension is missing: there is no CPU fallback."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RXHIP_LIB: another build of the SAME library (A/B measurements of kernel variants, scripts/ab_variants.sh); never a fallback
LIB_PATH = os.environ.get("RXHIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "csrc", "librxhip.so"))

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_u64_p = ctypes.POINTER(ctypes.c_uint64)

OK, ERR_BADARG, ERR_UNSUPPORTED, ERR_NOT_POSDEF, ERR_NONFINITE_FE, ERR_HIP, ERR_NO_DEVICE, ERR_STATE, ERR_RCCL = range(9)
LAYOUT_TIME_CHAIN, LAYOUT_CHAIN_TIME = 0, 1
VAR_Y, VAR_X, VAR_U = 0, 1, 2
(K_SEG_AGGREGATE, K_BOUNDARY_SCAN, K_FORWARD, K_BACKWARD, K_FE_REDUCE, K_GMM_PASS, K_GMM_REDUCE, K_GMM_UPDATE, K_HGF_FILTER,
 K_DRIFT_CHAIN, K_COUNT) = range(11)
KERNEL_NAMES = ["k_seg_aggregate", "k_boundary_scan", "k_forward", "k_backward", "k_fe_reduce", "k_gmm_pass", "k_gmm_reduce",
                "k_gmm_update", "k_hgf_filter", "k_drift_chain"]


class LgssmDesc(ctypes.Structure):
    _fields_ = [
        ("d", ctypes.c_int32), ("dy", ctypes.c_int32), ("T", ctypes.c_int64), ("n_chains", ctypes.c_int64),
        ("n_models", ctypes.c_int32), ("prior_through_transition", ctypes.c_int32),
        ("A", c_double_p), ("B", c_double_p), ("P", c_double_p), ("Q", c_double_p), ("m0", c_double_p),
        ("V0", c_double_p), ("chain_model", c_int32_p), ("segments", ctypes.c_int32), ("device", ctypes.c_int32),
        ("stream", ctypes.c_void_p), ("horizon", ctypes.c_int64), ("allow_missing", ctypes.c_int32),
        ("step_model", c_int32_p), ("state_offset", c_double_p), ("obs_offset", c_double_p),
    ]


c_int64_p = ctypes.POINTER(ctypes.c_int64)
VARKIND_RANDOM, VARKIND_DATA, VARKIND_CONST = 0, 1, 2
NODE_MVNORMAL_MEAN_COV, NODE_MULTIPLY = 1, 2
(NODE_NORMAL_MEAN_VARIANCE, NODE_NORMAL_MEAN_PRECISION, NODE_GAMMA_SHAPE_RATE, NODE_DIRICHLET, NODE_BETA, NODE_CATEGORICAL,
 NODE_BERNOULLI, NODE_NORMAL_MIXTURE, NODE_GCV, NODE_WISHART, NODE_ADD, NODE_MVNORMAL_MEAN_PRECISION, NODE_GAMMA_SHAPE_SCALE) = range(3, 16)
INIT_NONE, INIT_NORMAL, INIT_GAMMA, INIT_DIRICHLET, INIT_MVNORMAL, INIT_WISHART = 0, 1, 2, 3, 4, 5


class GraphDesc(ctypes.Structure):
    _fields_ = [("n_variables", ctypes.c_int64), ("var_kind", c_int32_p), ("var_rows", c_int32_p), ("var_cols", c_int32_p),
                ("var_const", c_int64_p), ("n_factors", ctypes.c_int64), ("factor_type", c_int32_p),
                ("factor_iface", c_int64_p), ("const_pool", c_double_p), ("n_const", ctypes.c_int64),
                ("n_replicas", ctypes.c_int64),
                # optional extensions (zero = 3-interface Gaussian graphs)
                ("factor_iface_ptr", c_int64_p), ("var_init_family", c_int32_p), ("var_init", c_int64_p),
                ("gh_points", ctypes.c_int32), ("n_observations", ctypes.c_int64), ("allow_missing", ctypes.c_int32),
                # the factorisation of q around every node (VariationalConstraintsFactorizationIndicesKey): NULL or one cluster id per factor_iface entry
                ("factor_cluster", c_int32_p)]


class LgssmLowered(ctypes.Structure):
    _fields_ = [("d", ctypes.c_int32), ("dy", ctypes.c_int32), ("T", ctypes.c_int64),
                ("prior_through_transition", ctypes.c_int32), ("A", c_double_p), ("B", c_double_p), ("P", c_double_p),
                ("Q", c_double_p), ("m0", c_double_p), ("V0", c_double_p), ("state_var", c_int64_p), ("data_var", c_int64_p),
                ("deterministic", ctypes.c_int32), ("c", c_double_p), ("n_models", ctypes.c_int32), ("step_model", c_int32_p),
                ("has_offsets", ctypes.c_int32), ("state_offset", c_double_p), ("obs_offset", c_double_p),
                ("du", ctypes.c_int32), ("input_matrix", c_double_p), ("input_var", c_int64_p)]


class LgssmNoiseLowered(ctypes.Structure):
    _fields_ = [("chain", LgssmLowered), ("precision_var", ctypes.c_int64), ("nu0", ctypes.c_double), ("init_nu", ctypes.c_double),
                ("S0", c_double_p), ("init_V", c_double_p)]


class GmmLowered(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int64), ("K", ctypes.c_int32)] + [(n, c_double_p) for n in (
        "mu0", "v0", "a0", "b0", "alpha0", "init_m_mean", "init_m_var", "init_p_shape", "init_p_rate", "init_s_alpha")] + [
        ("data_var", c_int64_p)]


class MvGmmLowered(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int64), ("K", ctypes.c_int32), ("d", ctypes.c_int32)] + [(n, c_double_p) for n in (
        "mu0", "S0", "nu0", "V0", "alpha0", "init_m_mean", "init_m_cov", "init_w_nu", "init_w_V", "init_s_alpha")] + [
        ("data_var", c_int64_p)]


class HgfLowered(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("kappa", "omega", "z_variance", "y_variance", "z0_mean", "z0_var", "x0_mean", "x0_var")] + [
        ("n_gh", ctypes.c_int32), ("zt_var", ctypes.c_int64), ("xt_var", ctypes.c_int64), ("y_var", ctypes.c_int64)]


class GmmDesc(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int64), ("K", ctypes.c_int32)] + [(n, c_double_p) for n in (
        "mu0", "v0", "a0", "b0", "alpha0", "init_m_mean", "init_m_var", "init_p_shape", "init_p_rate", "init_s_alpha")] + [
        ("materialize_responsibilities", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p)]


class MvGmmDesc(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int64), ("K", ctypes.c_int32), ("d", ctypes.c_int32)] + [(n, c_double_p) for n in (
        "mu0", "S0", "nu0", "V0", "alpha0", "init_m_mean", "init_m_cov", "init_w_nu", "init_w_V", "init_s_alpha")] + [
        ("materialize_responsibilities", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p)]


class DriftChainDesc(ctypes.Structure):
    _fields_ = [("T", ctypes.c_int64), ("n_chains", ctypes.c_int64), ("m0", ctypes.c_double), ("v0", ctypes.c_double),
                ("c", ctypes.c_double), ("obs_var", ctypes.c_double), ("prior_through_transition", ctypes.c_int32),
                ("device", ctypes.c_int32), ("stream", ctypes.c_void_p)]


class HgfDesc(ctypes.Structure):
    _fields_ = [("T", ctypes.c_int64), ("n_series", ctypes.c_int64)] + [(n, ctypes.c_double) for n in (
        "kappa", "omega", "z_variance", "y_variance", "z0_mean", "z0_var", "x0_mean", "x0_var")] + [
        ("n_gh", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p)]


class NoisePrior(ctypes.Structure):   # rxhip_noise_prior
    _fields_ = [("nu0", ctypes.c_double), ("S0", c_double_p), ("init_nu", ctypes.c_double), ("init_V", c_double_p)]


class TreeInfo(ctypes.Structure):   # rxhip_tree_info
    _fields_ = [("n_ops", ctypes.c_int64), ("n_levels", ctypes.c_int64), ("n_messages", ctypes.c_int64), ("doubles_per_replica", ctypes.c_int64),
                ("bytes_per_sweep", ctypes.c_int64), ("dmax", ctypes.c_int32), ("mode", ctypes.c_int32), ("replicas_per_workgroup", ctypes.c_int32),
                ("n_precision_vars", ctypes.c_int32), ("last_iteration_ms", ctypes.c_double), ("io_bytes_per_sweep", ctypes.c_int64),
                ("n_strands", ctypes.c_int64), ("n_strand_levels", ctypes.c_int64), ("longest_strand", ctypes.c_int32), ("kernels", ctypes.c_int32), ("strand_bytes_per_sweep", ctypes.c_int64), ("fe_bytes_per_sweep", ctypes.c_int64)]


class RuleCall(ctypes.Structure):   # rxhip_rule_call
    _fields_ = [("node_type", ctypes.c_int32), ("iface", ctypes.c_int32), ("d_out", ctypes.c_int32), ("d_in", ctypes.c_int32), ("n", ctypes.c_int64),
                ("constant", c_double_p), ("in_form", ctypes.c_int32), ("in_a", c_double_p), ("in_B", c_double_p), ("in2_a", c_double_p),
                ("in2_B", c_double_p), ("out_form", ctypes.c_int32), ("out_a", c_double_p), ("out_B", c_double_p)]


# every symbol include/rxhip.h declares: (name, restype, argtypes)
_H = ctypes.c_void_p
SYMBOLS = [
    ("rxhip_tree_create", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(_H)]),
    ("rxhip_tree_set_data", ctypes.c_int32, [_H, c_int64_p, ctypes.c_int64, c_double_p]),
    ("rxhip_tree_get_marginals", ctypes.c_int32, [_H, c_int64_p, ctypes.c_int64, c_double_p, c_double_p]),
    ("rxhip_tree_get_precision", ctypes.c_int32, [_H, ctypes.c_int64, c_double_p, c_double_p]),
    ("rxhip_tree_get_discrete", ctypes.c_int32, [_H, ctypes.c_int64, c_double_p, ctypes.POINTER(ctypes.c_int32)]),
    ("rxhip_tree_get_info", ctypes.c_int32, [_H, ctypes.POINTER(TreeInfo)]),
    ("rxhip_tree_continue", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_rule_eval", ctypes.c_int32, [ctypes.POINTER(RuleCall), ctypes.c_int32]),
    ("rxhip_lgssm_create", ctypes.c_int32, [ctypes.POINTER(LgssmDesc), ctypes.POINTER(_H)]),
    ("rxhip_lgssm_noise_create", ctypes.c_int32, [ctypes.POINTER(LgssmDesc), ctypes.POINTER(NoisePrior), ctypes.POINTER(_H)]),
    ("rxhip_lgssm_noise_get", ctypes.c_int32, [_H, c_double_p, c_double_p]),
    ("rxhip_lgssm_noise_continue", ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]),
    ("rxhip_graph_lower_lgssm", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.POINTER(LgssmLowered)]),
    ("rxhip_graph_lower_gmm", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.POINTER(GmmLowered)]),
    ("rxhip_graph_lower_mvgmm", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.POINTER(MvGmmLowered)]),
    ("rxhip_graph_lower_hgf", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.POINTER(HgfLowered)]),
    ("rxhip_graph_lower_lgssm_noise", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.POINTER(LgssmNoiseLowered)]),
    ("rxhip_lowering_error", ctypes.c_char_p, []),
    ("rxhip_lowering_asymmetry", ctypes.c_double, []),
    ("rxhip_tree_plan", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.POINTER(TreeInfo), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                         ctypes.POINTER(ctypes.c_uint64)]),
    ("rxhip_create", ctypes.c_int32, [ctypes.POINTER(GraphDesc), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.POINTER(_H)]),
    ("rxhip_lgssm_supported", ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32]),
    ("rxhip_set_data", ctypes.c_int32, [_H, ctypes.c_int32, c_double_p, ctypes.c_size_t, ctypes.c_int32]),
    ("rxhip_set_data_device", ctypes.c_int32, [_H, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32]),
    ("rxhip_run", ctypes.c_int32, [_H, ctypes.c_int32, ctypes.c_int32]),
    ("rxhip_run_async", ctypes.c_int32, [_H, ctypes.c_int32, ctypes.c_int32]),
    ("rxhip_run_filter", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_run_filter_async", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_sync", ctypes.c_int32, [_H]),
    ("rxhip_release_cached_memory", ctypes.c_int32, []),
    ("rxhip_set_caching", ctypes.c_int32, [ctypes.c_int32]),
    ("rxhip_set_conditioning_guard", ctypes.c_int32, [ctypes.c_int32]),
    ("rxhip_get_marginals", ctypes.c_int32, [_H, ctypes.c_int32, c_double_p, c_double_p, ctypes.c_int32]),
    ("rxhip_get_predictions", ctypes.c_int32, [_H, ctypes.c_int32, c_double_p, c_double_p, ctypes.c_int32]),
    ("rxhip_get_node_marginals", ctypes.c_int32, [_H, ctypes.c_int32, c_double_p, c_double_p, ctypes.c_int32]),
    ("rxhip_filter_step", ctypes.c_int32, [_H, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("rxhip_filter_reset", ctypes.c_int32, [_H]),
    ("rxhip_lgssm_set_offsets", ctypes.c_int32, [_H, c_double_p, c_double_p]),
    ("rxhip_lgssm_set_chain_offsets", ctypes.c_int32, [_H, c_double_p, c_double_p, ctypes.c_int32]),
    ("rxhip_lgssm_infer", ctypes.c_int32, [_H, c_double_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_double_p, c_double_p, c_double_p]),
    ("rxhip_get_marginals_device", ctypes.c_int32, [_H, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p),
                                                    ctypes.POINTER(ctypes.c_void_p)]),
    ("rxhip_get_marginals_chains", ctypes.c_int32, [_H, ctypes.c_int32, c_int64_p, ctypes.c_int64, c_double_p, c_double_p]),
    ("rxhip_comm_unique_id", ctypes.c_int32, [ctypes.c_char_p]),
    ("rxhip_comm_init_rank", ctypes.c_int32, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32,
                                              ctypes.c_int32]),
    ("rxhip_comm_destroy", ctypes.c_int32, [ctypes.c_void_p]),
    ("rxhip_comm_last_error", ctypes.c_char_p, []),
    ("rxhip_allreduce_free_energy", ctypes.c_int32, [_H, ctypes.c_void_p]),
    ("rxhip_gmm_allreduce_statistics", ctypes.c_int32, [_H, ctypes.c_void_p]),
    ("rxhip_get_free_energy", ctypes.c_int32, [_H, c_double_p]),
    ("rxhip_get_free_energy_per_chain", ctypes.c_int32, [_H, c_double_p]),
    ("rxhip_get_free_energy_device", ctypes.c_int32, [_H, ctypes.POINTER(ctypes.c_void_p)]),
    ("rxhip_copy_free_energy_to_device", ctypes.c_int32, [_H, ctypes.c_void_p]),
    ("rxhip_counters", ctypes.c_int32, [_H, c_u64_p, c_u64_p, c_u64_p]),
    ("rxhip_gmm_create", ctypes.c_int32, [ctypes.POINTER(GmmDesc), ctypes.POINTER(_H)]),
    ("rxhip_mvgmm_create", ctypes.c_int32, [ctypes.POINTER(MvGmmDesc), ctypes.POINTER(_H)]),
    ("rxhip_gmm_get_history", ctypes.c_int32, [_H, c_double_p]),
    ("rxhip_gmm_get_responsibilities", ctypes.c_int32, [_H, c_double_p]),
    ("rxhip_gmm_begin_run", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_gmm_accumulate", ctypes.c_int32, [_H]),
    ("rxhip_gmm_statistics_device", ctypes.c_int32, [_H, ctypes.POINTER(ctypes.c_void_p), c_int32_p]),
    ("rxhip_gmm_update", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_hgf_create", ctypes.c_int32, [ctypes.POINTER(HgfDesc), ctypes.POINTER(_H)]),
    ("rxhip_drift_chain_create", ctypes.c_int32, [ctypes.POINTER(DriftChainDesc), ctypes.POINTER(_H)]),
    ("rxhip_hgf_get_history", ctypes.c_int32, [_H, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_int32]),
    ("rxhip_set_profiling", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_get_kernel_times", ctypes.c_int32, [_H, c_double_p, c_u64_p]),
    ("rxhip_get_model_tables_ms", ctypes.c_int32, [_H, c_double_p]),
    ("rxhip_get_create_stages", ctypes.c_int32, [_H, c_double_p]),
    ("rxhip_set_covariance_mode", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_set_fixed_point_exits", ctypes.c_int32, [_H, ctypes.c_int32]),
    ("rxhip_reset_kernel_times", ctypes.c_int32, [_H]),
    ("rxhip_get_stream", ctypes.c_int32, [_H, ctypes.POINTER(ctypes.c_void_p)]),
    ("rxhip_get_schedule", ctypes.c_int32, [_H, c_int32_p, ctypes.POINTER(ctypes.c_int64)]),
    ("rxhip_last_error", ctypes.c_char_p, [_H]),
    ("rxhip_status_string", ctypes.c_char_p, [ctypes.c_int32]),
    ("rxhip_version", ctypes.c_char_p, []),
    ("rxhip_device_count", ctypes.c_int32, []),
    ("rxhip_destroy", ctypes.c_int32, [_H]),
]

_LIB = None


class RxHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rxhip status {status}: {message}")
        self.status = status


def lib():
    """Load librxhip.so.  Raises if the HIP extension has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C rxinfer.jl_amd/csrc)")
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
