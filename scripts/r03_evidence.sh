#!/bin/bash
# Round 3 evidence run (one gpurun call): the whole GPU suite, the default bench line, rocprofv3
# --kernel-trace --stats of the same command, the PMC passes behind profiles/hbm_traffic.json,
# BASELINE cfg3 / cfg4 under the profiler, and the headline frame compared with the reference over
# ALL 256 passes.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03z
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
tail -4 $OUT/pytest_gpu.log
# 1) the default line, as the driver runs it (N = 1, steps 1)
( timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 600 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
# 2) the same command under rocprofv3 (CPU legs and the child-process leg left out: they launch no kernels of this process)
cd /tmp && export TMPDIR=/tmp
P=$REPO/gpurun_out/prof_r03z
rm -rf $P; mkdir -p $P
CMD="python $REPO/bench.py --no-cpu-baseline --parity-passes 2 --no-strict"
echo "$CMD" > $P/command.txt
timeout 1200 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1
tail -c 400 $P/trace.log
# 3) PMC passes, 256 x 256 variant of the same workload (own runs, counters only)
CMD2="python $REPO/bench.py --width 256 --height 256 --steps 1 --no-cpu-baseline --no-parity"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/pmc1 -o pmc1 -- $CMD2 > $P/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/pmc2 -o pmc2 -- $CMD2 > $P/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $P/pmc3 -o pmc3 -- $CMD2 > $P/pmc3.log 2>&1
cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r03z gpurun_out/r03z/r03z_default > /dev/null 2>&1
# 4) BASELINE cfg3 / cfg4 lines, each under the profiler
cd /tmp
for c in cfg3 cfg4; do
  Q=$REPO/gpurun_out/prof_r03z_$c
  rm -rf $Q; mkdir -p $Q
  echo "python bench.py --config $c --no-cpu-baseline" > $Q/command.txt
  timeout 900 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config $c --no-cpu-baseline > $Q/trace.log 2>&1
  grep '^{' $Q/trace.log > $OUT/bench_$c.json
  ( cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r03z_$c gpurun_out/r03z/r03z_$c > /dev/null 2>&1 )
done
cd $REPO
ls -la $OUT
# 5) the metric's second half over ALL 256 passes (13 minutes of host work; SKIP_FULL_PARITY=1 skips it)
[ -n "$SKIP_FULL_PARITY" ] || ( timeout 2400 python bench.py --parity-passes 0 --no-cpu-baseline --no-secondary --no-other-configs --no-strict > $OUT/bench_full_parity.json 2> $OUT/bench_full_parity.err; echo "rc=$?" >> $OUT/bench_full_parity.err )
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r03z/bench_full_parity.json").read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ("value", "rmse_vs_ref", "max_abs_diff", "samples_word_count_differs", "samples", "pixels_bit_identical", "parity_passes", "reference_kind")})
except Exception as e:
    print("full parity:", e)
PY
