#!/bin/bash
# Round 3 evidence, part 2: BASELINE cfg3 / cfg4 lines under the profiler, with the parity leg on the
# same kernel variant as the timed render.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03z
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg3 cfg4; do
  Q=$REPO/gpurun_out/prof_r03z_$c
  rm -rf $Q; mkdir -p $Q
  echo "python bench.py --config $c --no-cpu-baseline" > $Q/command.txt
  timeout 900 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config $c --no-cpu-baseline > $Q/trace.log 2>&1
  grep '^{' $Q/trace.log > $OUT/bench_$c.json
  ( cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r03z_$c gpurun_out/r03z/r03z_$c > /dev/null 2>&1 )
  tail -c 300 $OUT/bench_$c.json
done
