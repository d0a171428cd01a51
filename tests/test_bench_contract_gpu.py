"""bench.py prints ONE JSON line with the contract's fields (small shape; the headline run is the default invocation)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_the_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--T", "3000",
                          "--chains", "128", "--cpu-sample-chains", "2", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    sys.path.insert(0, ROOT)
    import bench

    assert c["all_cores"]["cores"] == bench.host_cores()[0] and c["all_cores"]["value"] > 0  # the cgroup quota, not the visible threads
    assert c["free_energy_rel_vs_gpu"] < 1e-8  # both legs ran on the same observations
    ps = d["parity_spot"]
    assert ps["ok"] and ps["mean_rel"] < 1e-6 and ps["cov_rel"] < 1e-6 and ps["fe_rel"] < 1e-8


@pytest.mark.gpu
def test_bench_multi_gpu_request_on_a_single_gpu_box_fails_clearly():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run; with fewer than N
    devices it must say so instead of hanging in a rendezvous."""
    import torch

    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "HIP device" in (out.stderr + out.stdout)
