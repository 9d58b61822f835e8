#!/bin/bash
# round 6, last GPU call: BASELINE cfg4 rendered WHOLE (ce 2048 x 2048 @ 1024 spp: 4.3e9 samples) under the seed-matched
# policy with the final kernel (traceSequential<10,6,global,stack,2 masters,unit>), its sub-run rows [0, 32) on the same
# box, the byte comparison of the two - and the sha256 of the frame's fp64 sums against round 5's
# (profiles/r05w_cfg4_whole_frame_check.txt: 450f42c8...): the unit-level early-out must not have changed one bit.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06af; mkdir -p $O
timeout 300 python bench.py --config cfg4 --no-parity --no-cpu-baseline --no-secondary --dump-raw /tmp/ce_rows32.raw \
  > $O/bench_cfg4_rows32.json 2> $O/rows32.err; echo "sub-run rc=$?"
date +%s > $O/whole.start
timeout 2300 python bench.py --scene ce --width 2048 --height 2048 --spp 1024 --no-parity --no-cpu-baseline --no-secondary \
  --dump-raw /tmp/ce_whole.raw > $O/bench_cfg4_whole_frame.json 2> $O/whole.err; echo "whole frame rc=$?"
date +%s > $O/whole.end
python - <<PY 2>&1 | grep -v amdgpu.ids | tee $O/whole_frame_check.txt
import hashlib, json, sys
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as entry
pkg = entry.load_package()
line = [l for l in open("$O/bench_cfg4_whole_frame.json") if l.startswith("{")][-1]
d = json.loads(line)
sub = json.loads([l for l in open("$O/bench_cfg4_rows32.json") if l.startswith("{")][-1])
print("whole frame:", d["config"]["workload"], "| wall clock of the bench.py process by date(1): %d s" % (int(open("$O/whole.end").read()) - int(open("$O/whole.start").read())))
print("value %.4f Msamples/s, ms_per_step %.1f, kernel %s, frac %.4f, launches %d, avg_launch_ms %.1f, rays_per_sample %.3f"
      % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["launches"],
         d["roofline"]["avg_launch_ms"], d["roofline"]["rays_per_sample"]))
print("sub-run rows [0, 32) on the same box: %.4f Msamples/s, %s" % (sub["value"], sub["roofline"]["kernel"]))
rgb, cnt = pkg.raw_load("/tmp/ce_whole.raw")
srgb, scnt = pkg.raw_load("/tmp/ce_rows32.raw")
print("counts: every pixel %d samples: %s" % (cnt.flat[0], bool((cnt == 1024).all())))
print("rows [0, 32) of the whole frame == the sub-run's rows, bytes:", bool(np.array_equal(rgb[:32].view(np.uint64), srgb[:32].view(np.uint64))
      and np.array_equal(cnt[:32], scnt[:32])), "| sub-run rows [32, 2048) untouched:", bool((scnt[32:] == 0).all()))
print("finite:", bool(np.isfinite(rgb).all()), " mean radiance per channel:", (rgb.sum(axis=(0, 1)) / cnt.sum()).tolist())
h = hashlib.sha256(rgb.tobytes()).hexdigest()
print("sha256 of the fp64 sums:", h)
print("equal to round 5's whole frame (fused test, profiles/r05w_cfg4_whole_frame_check.txt):", h == "450f42c800a7aa109957ca625eb03ac75fc2fa96cef23af3c276daf2ba763e67")
PY
