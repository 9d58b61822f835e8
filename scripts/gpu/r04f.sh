#!/bin/bash
# round 4: the no-contraction build (make strict) against the reference's own code over ALL 256 passes of
# the headline frame - every pixel, every sample's RNG word count (11 minutes of host work).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04y; mkdir -p $O
PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_strict.so timeout 1500 python bench.py --parity-passes 0 --no-cpu-baseline --no-secondary --no-other-configs --no-strict > $O/bench_strict_all_256_passes.json 2> $O/bench_strict_all_256_passes.err; echo "rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04y/bench_strict_all_256_passes.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "rmse_vs_ref", "max_abs_diff", "samples_word_count_differs", "samples", "pixels_bit_identical", "pixels", "parity_passes", "reference_kind", "word_count_differences")})
PY
