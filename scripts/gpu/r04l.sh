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
# round 4: shares of the scene per worker wave by the wave's place - older / younger wave of a worker pair,
# master-side wave (PTW_SEQ_UNITS="o,y,m"; profiles/r04k_*: the younger waves are the slow ones) - on ce
# (rows [0, 8) of 2048 x 2048 @ 1024): parity of two settings against the oracle, then timing.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04l; mkdir -p $O
for u in "10,7,10" "9,8,10"; do
PTW_SEQ_UNITS=$u timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04l/summary.txt
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.getcwd() + "/tests")
import numpy as np, torch
import oracle_binding as ob
import test_gpu_round3 as r3
pkg = ob.pkg
cus = torch.cuda.get_device_properties(0).multi_processor_count
for name, edge, spp in (("ce", 5, cus + 1), ("ce", 4, 6)):
    if spp <= cus: os.environ["PTW_SEQ_MM"] = "1"
    scene = pkg.Scene(); cam = scene.build_named(name, edge, edge)
    params = pkg.default_params(width=edge, height=edge, samples_per_pixel=spp, seed=3)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=8)
    rgb, cnt, words, variant, _ = r3._render_with_stats(pkg, scene, cam, params)
    ok = np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words) and r3.rel_err(rgb, ref_rgb) < 1e-12
    print("PARITY units=%s %s %dx%d@%d %s: %s" % (os.environ["PTW_SEQ_UNITS"], name, edge, edge, spp, variant, "exact" if ok else "MISMATCH"))
PY
done
run() { local name=$1 units=$2
  PTW_SEQ_UNITS=$units timeout 300 python bench.py --scene ce --width 2048 --height 2048 --spp 1024 --rows 0:8 --no-cpu-baseline --no-parity --no-secondary > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" "$units" <<'PY' 2>&1 | tee -a gpurun_out/r04l/summary.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "units", sys.argv[3] or "default", "value %.3f" % r["value"], r["roofline"]["kernel"], "frac %.4f" % r["roofline"]["frac"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run ce_default ""
run ce_9_8_10 "9,8,10"
run ce_10_8_9 "10,8,9"
run ce_10_7_10 "10,7,10"
run ce_9_7_11 "9,7,11"
run ce_10_6_11 "10,6,11"
run ce_default_2 ""
