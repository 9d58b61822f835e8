# BASELINE config 5 (cornell 4096x4096 @ 4096 spp over 8 GPUs): the per-GPU share at N = 1, 2, 4, 8 GPUs
# (4096 / N passes of the full frame), timed on a stated sub-run: the first 16 rows of every pass
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02k; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "empty_shard" 2>&1 | tail -2
for n in 1 2 4 8; do
  spp=$((4096 / n))
  timeout 300 python bench.py --width 4096 --height 4096 --spp $spp --rows 0:16 --no-parity --no-cpu-baseline --no-secondary > $O/bench_cfg5_share_of_${n}gpus_${spp}passes_rows16.json 2>> $O/err.log
  python - <<PY
import json
d=json.load(open("$O/bench_cfg5_share_of_${n}gpus_${spp}passes_rows16.json"))
print("N=$n passes/GPU=$spp", round(d["value"],3), "Msamples/s per GPU", d["roofline"]["kernel"], "frac", round(d["roofline"]["frac"],4), "-> aggregate", round($n*d["value"],1))
PY
done
