#!/bin/bash
# Round 5: ONE evidence set, generated on the final tree in one gpurun call, everything under the prefix
# r05z: the whole GPU suite; the default bench line; bench.py --gpus 2 with two RCCL
# ranks on the one GPU (rccl_transport); rocprofv3 --kernel-trace --stats of the default command and of
# --config cfg3 (whole-frame parity x 2 passes) / cfg4 (rows [0, 64) x 2 passes); the PMC passes - SQ
# counters, FETCH_SIZE, WRITE_SIZE, each in its own run - of EVERY kernel name the driver's line carries
# (traceSequentialSpec and the per-pixel kernel at the 256 x 256 variant of the headline command, the
# lock-step per-pixel kernel at the BASELINE frame, <3,6,lds,2 masters> on suzanne, <10,6,global,2 masters>
# on ce): profiles/hbm_traffic.json is regenerated from them by scripts/update_hbm_traffic.py.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05z
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
[ -n "$SKIP_PYTEST" ] || ( timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -3
# 1) the default line, as the driver runs it at N = 1 (steps 1 here)
( timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 300 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
# 1b) N = 2 on this one GPU: bench.py launches its two ranks itself
( PTW_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --width 512 --height 512 --no-cpu-baseline --no-parity > $OUT/bench_two_ranks_one_gpu.json 2> $OUT/bench_two_ranks_one_gpu.err; echo "rc=$?" >> $OUT/bench_two_ranks_one_gpu.err )
tail -c 400 $OUT/bench_two_ranks_one_gpu.json; tail -2 $OUT/bench_two_ranks_one_gpu.err
# 2) the same command under rocprofv3 (CPU legs and the child-process leg left out: they launch no kernels of this process)
cd /tmp && export TMPDIR=/tmp
P=$REPO/gpurun_out/prof_r05z
rm -rf $P; mkdir -p $P
CMD="python $REPO/bench.py --no-cpu-baseline --parity-passes 2 --no-strict --no-other-configs"
echo "$CMD" > $P/command.txt
timeout 1200 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1
grep '^{' $P/trace.log > $OUT/bench_under_rocprof.json
tail -c 300 $P/trace.log
# 3) PMC passes, 256 x 256 variant of the same workload (own runs, counters only)
CMD2="python $REPO/bench.py --width 256 --height 256 --steps 1 --no-cpu-baseline --no-parity"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/pmc1 -o pmc1 -- $CMD2 > $P/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/pmc2 -o pmc2 -- $CMD2 > $P/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $P/pmc3 -o pmc3 -- $CMD2 > $P/pmc3.log 2>&1
cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r05z gpurun_out/r05z/r05z_default > /dev/null 2>&1
# 3b) the kernels of the driver's line, each alone (scripts/pmc_quick.sh: one counter set per run)
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
for sc in "cornell,256,256,256,0" "cornell,1024,1024,64,1,pix_kernel=1" "suzanne,256,256,512,0" "ce,64,64,1024,0"; do
  for c in "$SQ" "FETCH_SIZE" "WRITE_SIZE"; do
    echo "== $sc : $c"
    PMC="$c" bash scripts/pmc_quick.sh $sc 2>&1 | grep -v amdgpu.ids | grep "Msamples\|{" | tail -4
  done
done > $OUT/pmc_kernels.txt 2>&1
tail -12 $OUT/pmc_kernels.txt
# 4) BASELINE cfg3 / cfg4 lines, each under the profiler, with the wide parity windows
cd /tmp
for c in cfg3 cfg4; do
  Q=$REPO/gpurun_out/prof_r05z_$c
  rm -rf $Q; mkdir -p $Q
  if [ $c = cfg3 ]; then PAR="--parity-rows 1024 --parity-passes 2"; else PAR="--parity-rows 64 --parity-passes 2"; fi
  echo "python bench.py --config $c $PAR" > $Q/command.txt
  timeout 1500 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config $c $PAR > $Q/trace.log 2>&1
  grep '^{' $Q/trace.log > $OUT/bench_$c.json
  ( cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r05z_$c gpurun_out/r05z/r05z_$c > /dev/null 2>&1 )
done
cd $REPO
python - <<'PY'
import json
for name in ("bench_default", "bench_two_ranks_one_gpu", "bench_cfg3", "bench_cfg4"):
    try:
        r = json.loads(open(f"gpurun_out/r05z/{name}.json").read().strip().splitlines()[-1])
        keys = ("value", "n_gpus", "rccl_ranks", "value_tile_sharded", "rccl_transport", "rmse_vs_ref", "samples_word_count_differs", "picks_differ",
                "samples", "parity_rows", "parity_passes", "parity_kernel")
        print(name, {k: r.get(k) for k in keys if r.get(k) is not None}, "perpixel", (r.get("perpixel_policy") or {}).get("value"),
              "other", [(o.get("config"), o.get("value"), o.get("picks_differ"), (o.get("cpu_baseline") or {}).get("value")) for o in r.get("other_configs", [])],
              "strict", (r.get("strict_fp") or {}).get("value"), "cpu", (r.get("cpu_baseline") or {}).get("value"),
              "frac", r["roofline"]["frac"], "bytes", len(json.dumps(r)))
    except Exception as e:
        print(name, "FAILED:", e)
PY
