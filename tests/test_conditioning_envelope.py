"""The conditioning envelope of the information-form chain engines (csrc/model_envelope.hpp, include/rxhip.h rxhip_set_conditioning_guard): a model whose filtered
precisions are nearly singular — a vague prior in directions the observations do not see — is refused at creation BEFORE a device is touched (so the refusal itself is
tested on the CPU), and answered by the node-array executor when it arrives as a graph (GPU)."""
import ctypes

import numpy as np
import pytest

from rxhip import _lib, workloads


def _desc(mdl, T=5, C=1):
    d, dy = mdl["A"].shape[0], mdl["B"].shape[0]
    keep = [np.ascontiguousarray(mdl[k], dtype=np.float64) for k in ("A", "B", "P", "Q", "m0", "V0")]
    ds = _lib.LgssmDesc()
    ds.d, ds.dy, ds.T, ds.n_chains, ds.n_models = d, dy, T, C, 1
    for name, a in zip(("A", "B", "P", "Q", "m0", "V0"), keep):
        setattr(ds, name, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    ds.device = -1
    return ds, keep


def _create(mdl):
    L = _lib.lib()
    ds, keep = _desc(mdl)
    h = ctypes.c_void_p()
    st = L.rxhip_lgssm_create(ctypes.byref(ds), ctypes.byref(h))
    text = L.rxhip_lowering_error().decode()
    if h:
        L.rxhip_destroy(h)
    return st, bool(h), text


def _vague(d, dy, v0, q=1.0, seed=1):
    mdl = workloads.random_model(d, dy, seed=seed)
    mdl["V0"] = v0 * np.eye(d)
    mdl["Q"] = q * mdl["Q"]
    return mdl


@pytest.mark.parametrize("d,dy,v0,q,refused", [(32, 16, 1e4, 1e-2, True), (64, 32, 1e6, 1e-2, True), (32, 16, 1.0, 1.0, False), (32, 32, 1e2, 1.0, False),
                                                (16, 8, 1e2, 1.0, False), (16, 8, 1e8, 1e-2, True), (4, 2, 1e8, 1e-2, False)])
def test_refusal_comes_before_the_device_and_names_the_model(d, dy, v0, q, refused):
    st, handle, text = _create(_vague(d, dy, v0, q=q))
    if refused:
        assert st == _lib.ERR_UNSUPPORTED and not handle
        assert "kappa" in text and "node-array executor" in text and f"d = {d}" in text
    else:   # (inside the envelope, or not an information-form engine at all: creation goes on — to the device, which this machine may not have)
        assert st in (_lib.OK, _lib.ERR_NO_DEVICE)


def test_units_of_the_state_do_not_count():
    """x → S x with six decades between the components: A → S A S⁻¹, B → B S⁻¹, P → S P S, V0 → S V0 S describe the same model"""
    mdl = workloads.random_model(32, 16, seed=3)
    S = np.diag(10.0 ** np.linspace(-3, 3, 32))
    Si = np.linalg.inv(S)
    scaled = dict(A=S @ mdl["A"] @ Si, B=mdl["B"] @ Si, P=S @ mdl["P"] @ S, Q=mdl["Q"], m0=S @ mdl["m0"], V0=S @ mdl["V0"] @ S)
    assert _create(mdl)[0] in (_lib.OK, _lib.ERR_NO_DEVICE)
    assert _create(scaled)[0] in (_lib.OK, _lib.ERR_NO_DEVICE)


def test_the_check_can_be_switched_off():
    L = _lib.lib()
    mdl = _vague(32, 16, 1e4, q=1e-2)
    assert _create(mdl)[0] == _lib.ERR_UNSUPPORTED
    assert L.rxhip_set_conditioning_guard(0) == _lib.OK
    try:
        assert _create(mdl)[0] in (_lib.OK, _lib.ERR_NO_DEVICE)
    finally:
        L.rxhip_set_conditioning_guard(1)
    assert _create(mdl)[0] == _lib.ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("d,dy,v0", [(32, 16, 1e4), (64, 32, 1e4), (24, 7, 1e5)])
def test_a_refused_model_is_answered_by_the_executor(d, dy, v0):
    """through `infer` (which falls back as rxhip_create does) and through the graph entry point itself: posteriors and free energy against the oracle's
    Kalman / RTS restatement at the contract's bars — where the information-form engine, with the check off, is wrong by whole standard deviations"""
    import rxhip
    import rxoracle
    from rxhip import graph
    mdl = _vague(d, dy, v0, q=1e-2)
    T, C = 12, 2
    y = workloads.generate_batch(mdl, T, C, seed0=5, threads=1)   # [T][chain][dy]
    spec = rxhip.linear_gaussian_ssm(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    res = rxhip.infer(model=spec, data={"y": np.transpose(y, (1, 0, 2))}, free_energy=True)
    # the graph entry point: the pattern matcher's engine refuses, the executor takes the graph
    gb, xs, ys = graph.lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])[:3]
    from rxhip.tree import TreeEngine
    with TreeEngine(gb, n_replicas=C, force_executor=False) as te:
        assert te.info["dmax"] == d   # (an executor handle: the call reached rxhip_tree_create through rxhip_create)
        te.set_data(ys, np.transpose(y, (1, 0, 2)).reshape(C, T * dy))
        te.run(1, True)
        post = te.marginals(xs)
        tfe = te.free_energy_per_replica()
    for c in range(C):
        om, oc, onll = rxoracle.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, c])
        sd = np.sqrt(np.einsum("tii->ti", oc))
        for mean, cov, fe in ((res.posteriors["x"].mean[c], res.posteriors["x"].cov[c], res.free_energy[c][0]),
                              (np.stack([post[v][0][c] for v in xs]), np.stack([post[v][1][c] for v in xs]), tfe[c])):
            assert np.max(np.abs(mean - om) / sd) < 1e-6      # (the contract's bar: these models have κ up to 1e8)
            assert np.max(np.abs(cov - oc) / (sd[:, :, None] * sd[:, None, :])) < 1e-6
            assert abs(fe - onll) < 1e-9 * abs(onll)
    # what the check is there for
    L = _lib.lib()
    L.rxhip_set_conditioning_guard(0)
    try:
        try:
            with rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
                eng.set_data(y)
                eng.run(1, True)
                mean, _ = eng.marginals()
            om, oc, _ = rxoracle.lgssm_kalman_rts(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], y[:, 0])
            assert not np.max(np.abs(mean[:, 0] - om) / np.sqrt(np.einsum("tii->ti", oc))) < 1e-6
        except rxhip.RxHipError as err:   # (or its own recursions give up: a pivot that is not positive)
            assert err.status == _lib.ERR_NOT_POSDEF
    finally:
        L.rxhip_set_conditioning_guard(1)
