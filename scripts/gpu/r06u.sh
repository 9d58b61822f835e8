#!/bin/bash
# round 6, after the evidence set: (1) the small-scene part of the dispatch table with the two-wave form among the
# forced neighbours; (2) the headline frame with ALL 256 passes compared with the reference's own code (the kernel whose
# generator wave folds the samples; 15 minutes of host work).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06u; mkdir -p $out
SWEEP_SIZES=32,64 SWEEP_PASSES=256,384,512,768,1024 timeout 600 python scripts/dispatch_sweep.py $out/dispatch_sweep_small_scenes_three_kernels.md > $out/sweep.log 2>&1
tail -3 $out/sweep.log
( timeout 2400 python bench.py --parity-passes 0 --no-cpu-baseline --no-other-configs --no-strict --no-secondary > $out/bench_all_256_passes_parity.json 2> $out/bench_all_256_passes_parity.err; echo "rc=$?" >> $out/bench_all_256_passes_parity.err )
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06u/bench_all_256_passes_parity.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "rmse_vs_ref", "max_abs_diff", "samples_word_count_differs", "samples", "parity_passes", "pixels_bit_identical", "word_count_differences")})
PY
