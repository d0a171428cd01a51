"""The path's cross-GPU exchange on real RCCL.  A gpurun box has ONE GPU, and RCCL refuses two ranks on one device, so what
can run here is world size 1: communicator creation through the C ABI, the in-place exchange entry points on the engine's
stream, torch's `nccl` backend with `device_id=` through bench.py's N > 1 code path, and run-to-run bit-identity of the
free energy.  World size 2 is covered by the gloo tests (tests/test_distributed_cpu.py, test_gmm_gpu.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import rxhip
from rxhip import workloads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_communicator_and_free_energy_exchange():
    uid = rxhip.Communicator.unique_id()
    assert len(uid) == 128 and any(uid)
    mdl = workloads.c1_model()
    T, C = 500, 70
    y = workloads.generate_batch(mdl, T, C)
    with rxhip.Communicator(1, uid, 0) as comm, \
            rxhip.LGSSMEngine(mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"], T=T, n_chains=C) as eng:
        assert comm.handle
        eng.set_data(y)
        eng.run(3, True)
        fe0 = eng.free_energy()
        eng.allreduce_free_energy(comm)  # one rank: the sum over ranks is the local value, bit for bit
        eng.sync()
        assert np.array_equal(eng.free_energy(), fe0)
        eng.run(3, True)
        eng.allreduce_free_energy(comm)
        assert np.array_equal(eng.free_energy(), fe0)  # run-to-run identical
    with pytest.raises(rxhip.RxHipError):
        rxhip.Communicator(2, uid, 5)  # rank out of range


def test_c_abi_mixture_statistics_exchange():
    rng = np.random.default_rng(3)
    y = np.concatenate([rng.standard_normal(3000) - 5, rng.standard_normal(5000) + 4])
    priors = ([-4.0, 3.0], [1e2] * 2, [0.1] * 2, [0.1] * 2, [1.0] * 2)
    init = ([-4.0, 3.0], [1.0] * 2, [1.0] * 2, [1.0] * 2, [1.0] * 2)
    with rxhip.Communicator(1, rxhip.Communicator.unique_id(), 0) as comm, rxhip.GMMEngine(y.size, *priors, *init) as eng:
        eng.set_data(y)
        eng.run(5, True)
        h, f = eng.history(), eng.free_energy()
        eng.begin_run(5)
        for _ in range(5):
            eng.accumulate()
            eng.allreduce_statistics(comm)
            eng.update(True)
        eng.allreduce_free_energy(comm)
        eng.sync()
        assert np.array_equal(eng.history(), h) and np.array_equal(eng.free_energy(), f)


def _bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--T", "4000",
                          "--chains", "192", "--no-cpu-baseline", "--no-extras", *extra], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_torch_nccl_backend_through_the_bench_exchange_path():
    """bench.py --force-dist: init_process_group("nccl", device_id=…) at world size 1, the per-step all-gather of the
    free-energy scalar on the engine's stream + the ordered sum — the code an 8-GPU run executes."""
    a = _bench("--force-dist")
    b = _bench("--force-dist")
    c = _bench()
    assert a["parity_spot"]["ok"]
    assert a["free_energy_global"] == a["free_energy_rank0"] == b["free_energy_global"] == c["free_energy_rank0"]  # bit-identical


def test_c_abi_free_energy_exchange_of_the_node_array_executor():
    """replicas of an executor engine shard over ranks; the only exchange is the free-energy sum (one rank here: the local values, bit for bit)"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tree_graphs as tg
    from rxhip.tree import TreeEngine
    gb, ys, _ = tg.two_branch_chain(T=6)
    data = tg.random_data(gb, ys, 9, 3)
    with rxhip.Communicator(1, rxhip.Communicator.unique_id(), 0) as comm, TreeEngine(gb, n_replicas=9) as eng:
        with pytest.raises(rxhip.RxHipError):
            eng.allreduce_free_energy(comm)       # nothing has run yet
        eng.set_data(ys, data)
        eng.run(2, True)
        fe0 = eng.free_energy()
        eng.allreduce_free_energy(comm)
        assert np.array_equal(eng.free_energy(), fe0)
        eng.run(2, False)
        with pytest.raises(rxhip.RxHipError):
            eng.allreduce_free_energy(comm)       # the last run did not compute it

