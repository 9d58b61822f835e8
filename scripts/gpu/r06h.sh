#!/bin/bash
# round 6: the two-wave form of the speculative kernel (frontier + one candidate, two workgroups per CU) for scenes of
# at most 64 triangles with more passes than CUs: parity, then cfg5's per-GPU shares on one GPU (Cornell 4096 wide,
# rows [0, 16)) at 512 / 768 / 1024 passes under the three small-scene kernels, and the calibrated dispatcher.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06h; mkdir -p $out
python -m pytest tests/test_gpu_cli.py tests/test_gpu_round3.py tests/test_gpu_round6.py -x -q -m gpu -k "sequential_kernel_variants or small_scene" > $out/parity.log 2>&1
tail -5 $out/parity.log
for spp in 384 512 768 1024; do
  python scripts/quick_bench.py cornell,4096,16,$spp,0,seq_small_kernel=1 cornell,4096,16,$spp,0,seq_small_kernel=2 cornell,4096,16,$spp,0,seq_small_kernel=4
done > $out/small_scene_kernels_cfg5_shares.txt 2>&1
cat $out/small_scene_kernels_cfg5_shares.txt
python bench.py --width 4096 --height 4096 --spp 512 --rows 0:16 --no-cpu-baseline --no-parity --no-secondary > $out/bench_cfg5_share_512_passes.json 2> $out/bench_cfg5_share.err
tail -c 600 $out/bench_cfg5_share_512_passes.json | head -c 600; tail -3 $out/bench_cfg5_share.err
# ... and the default line once more on the final tree (the strict library rebuilt: r06_final.sh ran against a stale one)
mkdir -p gpurun_out/r06z
( timeout 1700 python bench.py > gpurun_out/r06z/bench_default.json 2> gpurun_out/r06z/bench_default.err; echo "rc=$?" >> gpurun_out/r06z/bench_default.err )
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06z/bench_default.json").read().strip().splitlines()[-1])
print("default", r["value"], "frac", r["roofline"]["frac"], "strict", r.get("strict_fp"), "bytes", len(json.dumps(r)))
PY
