"""bench.py's contract, on the GPU box: the one JSON line, plain and under torch.distributed.run
(one rank, so the RCCL init / barrier / framebuffer reduce path runs on a 1-GPU box too)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

SMALL = ["--width", "64", "--height", "64", "--spp", "16", "--steps", "2", "--warmup", "1"]
CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                 "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _json_line(proc):
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    return json.loads(lines[0])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_single_process_line():
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *SMALL, "--cpu-threads", "2",
                           "--cpu-frame", "96"],
                          capture_output=True, text=True, timeout=900, cwd=ROOT)
    r = _json_line(proc)
    assert CONTRACT_KEYS <= set(r)
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["unit"] == "Msamples/s"
    assert r["value"] > 0 and r["dtype"] == "f64" and r["vs_baseline"] is None
    # value is consistent with ms_per_step: 64*64*16 samples per step
    assert abs(r["value"] - 64 * 64 * 16 / (r["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * r["value"] + 1e-9
    roof = r["roofline"]
    assert roof["kernel"].startswith("traceSequential") and roof["launches"] == 2
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert 20 < roof["rays_per_sample"] < 80
    cpu = r["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["cores"] == 2 and cpu["kind"] in ("reference", "port")
    assert [leg["cores"] for leg in r["cpu_baseline_legs"]][:2] == [1, 2]
    assert r["perpixel_policy"]["value"] > r["value"]
    assert r["end_to_end_ms_per_step"] > r["ms_per_step"]
    # the metric's second half: the whole frame against the reference's own code
    if r.get("rmse_vs_ref") is not None:
        # (the comparison is with the reference built with ITS flags, -funsafe-math-optimizations:
        # the last bits of a sum may differ where either compiler contracted a*b+c)
        assert max(r["rmse_vs_ref"]) < 1e-12 and r["max_abs_diff"] < 1e-12
        assert r["samples_word_count_differs"] == 0 and r["counts_equal"]
        assert r["pixels_bit_identical"] >= 0.5 * r["pixels"]
        assert r["cpu_baseline_legs"][-1]["cores"] >= 1


def test_bench_under_torchrun_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "1", *SMALL, "--no-cpu-baseline"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    r = _json_line(proc)
    assert CONTRACT_KEYS <= set(r)
    assert r["n_gpus"] == 1 and r["scaling"] == "strong" and r["value"] > 0


def test_bench_weak_flag_and_perpixel_policy():
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *SMALL, "--scaling", "weak", "--policy",
                           "perpixel", "--no-cpu-baseline", "--no-parity"],
                          capture_output=True, text=True, timeout=900, cwd=ROOT)
    r = _json_line(proc)
    assert r["scaling"] == "weak" and r["roofline"]["kernel"] == "tracePerPixelPersistent"
