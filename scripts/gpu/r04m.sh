#!/bin/bash
# round 4, last GPU call: the shares by the wave's place (seqUnitSplitByPlace) as shipped - parity of
# everything two-master, then BASELINE cfg4 under rocprofv3 with the parity window rows [0, 64) x 2 passes.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$PWD
O=gpurun_out/r04w; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_round3.py tests/test_gpu_cli.py tests/test_gpu_round4.py -x -q -m gpu -k "two_master or ties or decoupled" > $O/pytest_two_master.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_two_master.log
cd /tmp
Q=$REPO/gpurun_out/prof_r04w_cfg4
rm -rf $Q; mkdir -p $Q
echo "python bench.py --config cfg4 --no-cpu-baseline --parity-rows 64 --parity-passes 2" > $Q/command.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config cfg4 --no-cpu-baseline --parity-rows 64 --parity-passes 2 > $Q/trace.log 2>&1
grep '^{' $Q/trace.log > $REPO/$O/bench_cfg4.json
cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r04w_cfg4 gpurun_out/r04w/r04w_cfg4 > /dev/null 2>&1
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04w/bench_cfg4.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step", "rmse_vs_ref", "samples_word_count_differs", "samples", "parity_rows", "parity_passes", "parity_kernel")}, r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"])
PY
head -16 gpurun_out/r04w/r04w_cfg4_rocprof_summary.md | tail -8
