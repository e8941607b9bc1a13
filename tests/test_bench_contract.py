"""bench.py's output contract (the driver parses this line): one JSON object with the metric, the workload, the
roofline of the dominant kernel family and the CPU baseline measured beside it."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_json_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "2", "--batch", "64", "--cpu-steps", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


@pytest.mark.gpu
def test_bench_two_rank_rehearsal():
    """The driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` on an 8-GPU node.  On the
    one-GPU test box the same launch line runs with MMFM_BENCH_REHEARSE=1 (both ranks on cuda:0, gloo): one JSON line from
    rank 0, whole-job value = 2 ranks x batch / max-over-ranks time."""
    env = dict(os.environ, MMFM_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "32", "--no-kernel-profile"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
