# rocprofv3 --kernel-trace --stats of the default bench command and of the suzanne full-size command, end of round 2
export REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in cornell suzanne; do
  O=$REPO/gpurun_out/prof_r02w_$tag; rm -rf $O; mkdir -p $O/trace
  if [ $tag = cornell ]; then ARGS="--no-cpu-baseline --no-parity"; else ARGS="--scene suzanne --spp 512 --no-cpu-baseline --no-parity"; fi
  echo "python bench.py $ARGS  (rocprofv3 --kernel-trace --stats)" > $O/command.txt
  rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $REPO/bench.py $ARGS > $O/bench.json 2> $O/log.txt
  find $O/trace -name "*.db" | head -2; tail -c 600 $O/bench.json | head -c 300; echo
  cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r02w_$tag gpurun_out/r02w_${tag}_rocprof_summary; cd /tmp
  python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("$tag", d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['perpixel_policy']['value'], d['perpixel_policy']['roofline']['avg_launch_ms'])
PY
done
cat $REPO/gpurun_out/r02w_*_rocprof_summary.md | head -40
