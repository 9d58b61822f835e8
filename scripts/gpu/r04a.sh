#!/bin/bash
# RETIRED (round 6): this script drives the library through the PTW_SEQ_* / PTW_PIX_* / PTW_TEST_*
# environment switches of rounds 2-4.  ABI v5 (commit 916a1dc is the last with them) replaced those by
# ptw_debug_options (`--debug name=value` in the CLI and bench.py, Context.set_debug in Python): run against
# HEAD it would time the DEFAULT dispatch under the old labels.  Kept as the record of how profiles/r04* were
# taken; to re-run it, check out 916a1dc.
if [ "${PTW_ALLOW_RETIRED_SCRIPT:-0}" != "1" ]; then
  echo "$0: retired - needs commit 916a1dc (the PTW_SEQ_*/PTW_PIX_* environment switches are gone; use --debug)" >&2
  exit 2
fi
# round 4, first GPU call: the new multi-GPU-on-one-GPU tests, the explicit kernel choice, the
# PERPIXEL ray counter, lane occupancy of the lock-step kernel, the exact-decisions build.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_bench.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_new.log
timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "perpixel_kernel_choice or calibration or failing_shard or render_ex or loopback or rccl" > $O/pytest_r3.log 2>&1; echo "pytest_r3 rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_r3.log
timeout 300 python bench.py --policy perpixel --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench_perpixel.json 2> $O/bench_perpixel.err; echo "bench_perpixel rc=$?" | tee -a $O/summary.txt
python - <<'PY' 2>&1 | tee -a gpurun_out/r04a/summary.txt
import json
r = json.loads(open("gpurun_out/r04a/bench_perpixel.json").read().strip().splitlines()[-1])
print("perpixel cornell 1024@256:", r["value"], r["roofline"]["kernel"], r["roofline"]["frac"], "line bytes", len(json.dumps(r)))
PY
# lane occupancy of the lock-step kernel (prof build)
for s in 0 1; do
PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_prof.so PTW_PIX_COUNT_SLOTS=$s timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04a/summary.txt
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as e
pkg = e.load_package()
for name in ("cornell", "suzanne"):
    w = h = 512; spp = 32
    scene = pkg.Scene(); cam = scene.build_named(name, w, h)
    ctx = pkg.Context(0); ctx.set_scene(scene); ctx.enable_stats(True)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda"); cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    p = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1, rng_policy=1, pix_kernel=pkg.PIX_KERNEL_LOCKSTEP)
    ctx.render(cam, p, rgb.data_ptr(), cnt.data_ptr()); torch.cuda.synchronize()
    st = ctx.stats(reset=True)
    print("COUNT_SLOTS=%s" % os.environ["PTW_PIX_COUNT_SLOTS"], name, st.trace_kernel.decode(), "counter per sample: %.3f" % (st.rays / st.samples), "ms %.1f" % st.trace_ms)
PY
done
timeout 600 python scripts/exact_words_probe.py libptw_hip_exact.so libptw_hip.so 2>&1 | grep -v amdgpu.ids | tee $O/exact_words_probe.txt
timeout 400 python bench.py --config cfg3 --no-cpu-baseline --no-parity > $O/bench_cfg3_base.json 2> $O/bench_cfg3_base.err; echo "cfg3 rc=$?" | tee -a $O/summary.txt
python - <<'PY' 2>&1 | tee -a gpurun_out/r04a/summary.txt
import json
r = json.loads(open("gpurun_out/r04a/bench_cfg3_base.json").read().strip().splitlines()[-1])
print("cfg3 base:", r["value"], r["roofline"]["kernel"], r["roofline"]["frac"])
PY
