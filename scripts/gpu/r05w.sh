#!/bin/bash
# round 5, last GPU call: (1) pytest -m gpu on the final tree; (2) BASELINE cfg4 rendered WHOLE under the
# seed-matched policy (2048 x 2048 @ 1024 spp: 4.3e9 samples, about 34 minutes) - every earlier cfg4 number is
# the stated prefix sub-run rows [0, 32); (3) the property that makes the sub-run a fair stand-in: rows [0, 32)
# of the whole frame are byte for byte the sub-run's rows.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${1:-r05w}; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python bench.py --config cfg4 --no-parity --no-cpu-baseline --no-secondary --dump-raw /tmp/ce_rows32.raw \
  > $O/bench_cfg4_rows32.json 2> $O/rows32.err; echo "sub-run rc=$?"
date +%s > $O/whole.start
timeout 2500 python bench.py --scene ce --width 2048 --height 2048 --spp 1024 --no-parity --no-cpu-baseline --no-secondary \
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
print("whole frame:", d["config"]["workload"])
print("value %.4f Msamples/s, ms_per_step %.1f, kernel %s, frac %.4f, launches %d, avg_launch_ms %.1f, rays_per_sample %.3f"
      % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["launches"],
         d["roofline"]["avg_launch_ms"], d["roofline"]["rays_per_sample"]))
rgb, cnt = pkg.raw_load("/tmp/ce_whole.raw")
srgb, scnt = pkg.raw_load("/tmp/ce_rows32.raw")
print("counts: every pixel %d samples: %s" % (cnt.flat[0], bool((cnt == 1024).all())))
print("rows [0, 32) of the whole frame == the sub-run's rows, bytes:", bool(np.array_equal(rgb[:32].view(np.uint64), srgb[:32].view(np.uint64))
      and np.array_equal(cnt[:32], scnt[:32])), "| sub-run rows [32, 2048) untouched:", bool((scnt[32:] == 0).all()))
print("finite:", bool(np.isfinite(rgb).all()), " mean radiance per channel:", (rgb.sum(axis=(0, 1)) / cnt.sum()).tolist())
print("sha256 of the fp64 sums:", hashlib.sha256(rgb.tobytes()).hexdigest())
small_rgb = rgb.reshape(512, 4, 512, 4, 3).sum(axis=(1, 3))
small_cnt = cnt.reshape(512, 4, 512, 4).sum(axis=(1, 3)).astype(np.uint32)
pkg.png_save("$O/ce_2048x2048_1024spp_sequential_downsampled_512.png", pkg.pixels_rgb8(small_rgb, small_cnt))
PY
