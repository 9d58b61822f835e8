#!/bin/bash
# Round 6, third session: the evidence set of the FINAL tree (after the shares by place from 12 / 8 units, the <9,7> /
# <10,7> instantiations and the unit-level u-first early-out), one gpurun call, prefix r06zz.  What the third session
# did not touch (the PERPIXEL kernels, the accelerated modes) keeps its r06z files; everything the driver's line carries
# is measured again here: the whole GPU suite, the default bench line, rocprofv3 of the default command and of --config
# cfg3 / cfg4 with the wide parity windows, the PMC passes of the three kernels the line names, two ranks on the one
# GPU, the dispatch sweep of the worker-wave sizes.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06zz
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests -q -m gpu --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -3
( timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 300 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
( PTW_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --width 512 --height 512 --no-cpu-baseline --no-parity > $OUT/bench_two_ranks_one_gpu.json 2> $OUT/bench_two_ranks_one_gpu.err; echo "rc=$?" >> $OUT/bench_two_ranks_one_gpu.err )
cd /tmp && export TMPDIR=/tmp
P=$REPO/gpurun_out/prof_r06zz
rm -rf $P; mkdir -p $P
CMD="python $REPO/bench.py --no-cpu-baseline --parity-passes 2 --no-strict --no-other-configs"
echo "$CMD" > $P/command.txt
timeout 1200 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1
grep '^{' $P/trace.log > $OUT/bench_under_rocprof.json
CMD2="python $REPO/bench.py --width 256 --height 256 --steps 1 --no-cpu-baseline --no-parity --no-other-configs --no-strict"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/pmc1 -o pmc1 -- $CMD2 > $P/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/pmc2 -o pmc2 -- $CMD2 > $P/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $P/pmc3 -o pmc3 -- $CMD2 > $P/pmc3.log 2>&1
cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r06zz gpurun_out/r06zz/r06zz_default > /dev/null 2>&1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
for sc in "suzanne,256,256,512,0" "ce,64,64,1024,0" "ce,64,64,1024,0,seq_unit_ufirst=0" "ce,128,64,256,0"; do
  for c in "$SQ" "FETCH_SIZE" "WRITE_SIZE"; do
    echo "== $sc : $c"
    PMC="$c" bash scripts/pmc_quick.sh $sc 2>&1 | grep -v amdgpu.ids | grep "Msamples\|{" | tail -4
  done
done > $OUT/pmc_kernels.txt 2>&1
tail -4 $OUT/pmc_kernels.txt
cd /tmp
for c in cfg3 cfg4; do
  Q=$REPO/gpurun_out/prof_r06zz_$c
  rm -rf $Q; mkdir -p $Q
  if [ $c = cfg3 ]; then PAR="--parity-rows 1024 --parity-passes 2"; else PAR="--parity-rows 64 --parity-passes 2"; fi
  echo "python bench.py --config $c $PAR" > $Q/command.txt
  timeout 1500 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config $c $PAR > $Q/trace.log 2>&1
  grep '^{' $Q/trace.log > $OUT/bench_$c.json
  ( cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r06zz_$c gpurun_out/r06zz/r06zz_$c > /dev/null 2>&1 )
done
cd $REPO
SWEEP_SIZES=256,512,1000,1900,3400,8192 SWEEP_PASSES=256,512,1024 SWEEP_POLICIES=0 timeout 600 python scripts/dispatch_sweep.py $OUT/dispatch_sweep_worker_wave_sizes.md > $OUT/dispatch_sweep.log 2>&1
tail -3 $OUT/dispatch_sweep.log
python - <<'PY'
import json
for name in ("bench_default", "bench_under_rocprof", "bench_two_ranks_one_gpu", "bench_cfg3", "bench_cfg4"):
    try:
        r = json.loads(open(f"gpurun_out/r06zz/{name}.json").read().strip().splitlines()[-1])
        keys = ("value", "n_gpus", "rccl_ranks", "value_tile_sharded", "rmse_vs_ref", "samples_word_count_differs", "picks_differ",
                "samples", "parity_rows", "parity_passes", "parity_kernel")
        print(name, {k: r.get(k) for k in keys if r.get(k) is not None}, "perpixel", (r.get("perpixel_policy") or {}).get("value"),
              "other", [(o.get("config"), o.get("value"), o.get("picks_differ"), (o.get("cpu_baseline") or {}).get("value"), o.get("prefilter_mode")) for o in r.get("other_configs", [])],
              "strict", (r.get("strict_fp") or {}).get("value"), "cpu", (r.get("cpu_baseline") or {}).get("value"),
              "kernel", r["roofline"]["kernel"], "frac", r["roofline"]["frac"], "bytes", len(json.dumps(r)))
    except Exception as e:
        print(name, "FAILED:", e)
PY
