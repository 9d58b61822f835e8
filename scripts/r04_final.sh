#!/bin/bash
# Round 4, last evidence run on the library AS SHIPPED (after the pick / LDS-min change of the
# worker-wave kernels, which leaves traceSequentialSpec and the PERPIXEL kernels byte for byte as
# scripts/r04_evidence.sh measured them): the whole GPU suite, the default bench line (its
# other_configs are the changed kernels), cfg3 / cfg4 under rocprofv3 with the wide parity windows, and
# SQ / traffic counters of the two-master kernels.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04y
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
tail -4 $OUT/pytest_gpu.log
( timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 200 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
for c in cfg3 cfg4; do
  Q=$REPO/gpurun_out/prof_r04y_$c
  rm -rf $Q; mkdir -p $Q
  if [ $c = cfg3 ]; then PAR="--parity-rows 1024 --parity-passes 2"; else PAR="--parity-rows 64 --parity-passes 2"; fi
  echo "python bench.py --config $c --no-cpu-baseline $PAR" > $Q/command.txt
  timeout 1500 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config $c --no-cpu-baseline $PAR > $Q/trace.log 2>&1
  grep '^{' $Q/trace.log > $OUT/bench_$c.json
  ( cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r04y_$c gpurun_out/r04y/r04y_$c > /dev/null 2>&1 )
done
cd $REPO
# SQ / HBM counters of the two-master kernels (own --pmc runs, counters only)
for sc in "suzanne,256,256,512,0" "ce,64,64,1024,0"; do
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    echo "== $sc : $c"
    PMC="$c" bash scripts/pmc_quick.sh $sc 2>&1 | grep -v amdgpu.ids | tail -6
  done
done > $OUT/pmc_two_master.txt 2>&1
tail -30 $OUT/pmc_two_master.txt
python - <<'PY'
import json
for name in ("bench_default", "bench_cfg3", "bench_cfg4"):
    try:
        r = json.loads(open(f"gpurun_out/r04y/{name}.json").read().strip().splitlines()[-1])
        keys = ("value", "rmse_vs_ref", "samples_word_count_differs", "samples", "parity_rows", "parity_passes", "parity_kernel")
        print(name, {k: r.get(k) for k in keys}, "perpixel", (r.get("perpixel_policy") or {}).get("value"),
              "other", [(o.get("config"), o.get("value"), o.get("frac")) for o in r.get("other_configs", [])], "strict", (r.get("strict_fp") or {}).get("value"),
              "frac", r["roofline"]["frac"], "bytes", len(json.dumps(r)))
    except Exception as e:
        print(name, "FAILED:", e)
PY
