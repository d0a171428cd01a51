// wave_diff.cpp — test harness: runs an op table through the register rule bodies (tree_kernels.hpp, template dimension N) or through the LDS-staged ones
// (tree_wave_kernels.hpp, RXHIP_HOST_EMUL: a wavefront of one lane) on host arrays.  Built by tests/test_tree_wave_host.py with g++.
#include "tree_wave_kernels.hpp"

using namespace rxhip::tree;

template <int N>
static void run_lane(const TreeParams& p, int n_ops) {
    for (int o = 0; o < n_ops; ++o) {
        const int* w = p.ops + (size_t)o * OP_WORDS;
        for (long long r = 0; r < p.R; ++r) {
            if (w[W_OP] <= OP_MARGINAL) eval_bp<N>(p, w, r);
            else eval_fe<N>(p, w, r);
        }
    }
}
static void run_wave(const TreeParams& p, int n_ops, int dmax) {
    const wave::Ctx c = wave::make_ctx(dmax);
    for (int o = 0; o < n_ops; ++o) {
        const int* w = p.ops + (size_t)o * OP_WORDS;
        for (long long r = 0; r < p.R; ++r) {
            if (w[W_OP] <= OP_MARGINAL) wave::eval_bp<64>(c, p, w, r);
            else wave::eval_fe<64>(c, p, w, r);
        }
    }
}

extern "C" int emul_run(int which, int n, const int* ops, int n_ops, const int* aux, const double* cpool, double* msg, double* marg, double* val, double* prec,
                        double* term, double* stat, long long R, long long RS, int want_fe) {
    int status = 0;
    TreeParams p{};
    p.ops = ops; p.aux = aux; p.cpool = cpool; p.msg = msg; p.marg = marg; p.val = val; p.prec = prec; p.term = term; p.stat = stat;
    p.R = R; p.RS = RS; p.want_fe = want_fe; p.status = &status;
    p.es = RS; p.rs_msg = p.rs_marg = p.rs_val = p.rs_prec = p.rs_term = p.rs_stat = 1;   // the replica-fastest layout both bodies can read
    if (which == 1) run_wave(p, n_ops, n);
    else if (n <= 8) run_lane<8>(p, n_ops);
    else if (n <= 16) run_lane<16>(p, n_ops);
    else if (n <= 32) run_lane<32>(p, n_ops);
    else run_lane<64>(p, n_ops);
    return status;
}
