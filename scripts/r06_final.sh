#!/bin/bash
# Round 6: ONE evidence set, generated on the final tree in one gpurun call, everything under the prefix r06z:
# the whole GPU suite; the default bench line; bench.py --gpus 2 with two RCCL ranks on the one GPU; rocprofv3
# --kernel-trace --stats of the default command and of --config cfg3 / cfg4; the PMC passes (SQ counters, FETCH_SIZE,
# WRITE_SIZE, each in its own run) of every kernel name the driver's line carries plus the prefilter kernels;
# the separate prefilter mode's own bench lines; the dispatch sweep on the final dispatcher.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06z
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
[ -n "$SKIP_PYTEST" ] || ( timeout 1700 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -3
# 1) the default line, as the driver runs it at N = 1 (steps 1 here)
( timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 300 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
# 1b) N = 2 on this one GPU: bench.py launches its two ranks itself
( PTW_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --width 512 --height 512 --no-cpu-baseline --no-parity > $OUT/bench_two_ranks_one_gpu.json 2> $OUT/bench_two_ranks_one_gpu.err; echo "rc=$?" >> $OUT/bench_two_ranks_one_gpu.err )
tail -c 300 $OUT/bench_two_ranks_one_gpu.json; tail -2 $OUT/bench_two_ranks_one_gpu.err
# 1c) the separate prefilter mode's own lines (perpixel policy: suzanne, ce; sequential policy: cfg4's sub-run)
( timeout 600 python bench.py --scene suzanne --spp 64 --policy perpixel --accel prefilter --no-cpu-baseline --no-parity --no-secondary > $OUT/bench_prefilter_suzanne_perpixel.json 2> $OUT/bench_prefilter_suzanne_perpixel.err )
( timeout 600 python bench.py --scene ce --width 512 --height 512 --spp 64 --policy perpixel --accel prefilter --no-cpu-baseline --no-parity --no-secondary > $OUT/bench_prefilter_ce_perpixel.json 2> $OUT/bench_prefilter_ce_perpixel.err )
( timeout 900 python bench.py --config cfg4 --accel prefilter --no-cpu-baseline --parity-passes 2 > $OUT/bench_prefilter_cfg4_sequential.json 2> $OUT/bench_prefilter_cfg4_sequential.err )
tail -c 200 $OUT/bench_prefilter_cfg4_sequential.json
# 2) the default command under rocprofv3 (CPU legs and the child-process leg left out: they launch no kernels of this process)
cd /tmp && export TMPDIR=/tmp
P=$REPO/gpurun_out/prof_r06z
rm -rf $P; mkdir -p $P
CMD="python $REPO/bench.py --no-cpu-baseline --parity-passes 2 --no-strict --no-other-configs"
echo "$CMD" > $P/command.txt
timeout 1200 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1
grep '^{' $P/trace.log > $OUT/bench_under_rocprof.json
tail -c 300 $P/trace.log
# 3) PMC passes, 256 x 256 variant of the same workload (own runs, counters only)
CMD2="python $REPO/bench.py --width 256 --height 256 --steps 1 --no-cpu-baseline --no-parity --no-other-configs --no-strict"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/pmc1 -o pmc1 -- $CMD2 > $P/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/pmc2 -o pmc2 -- $CMD2 > $P/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $P/pmc3 -o pmc3 -- $CMD2 > $P/pmc3.log 2>&1
cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r06z gpurun_out/r06z/r06z_default > /dev/null 2>&1
# 3b) the kernels of the driver's line and the prefilter kernels, each alone (scripts/pmc_quick.sh: one counter set per run)
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
for sc in "cornell,256,256,256,0" "cornell,1024,1024,64,1,pix_kernel=1" "suzanne,256,256,512,0" "ce,64,64,1024,0" \
          "ce,64,64,1024,0,accel=2" "ce,256,256,16,1" "ce,256,256,16,1,accel=2" "suzanne,512,512,16,1,accel=2"; do
  for c in "$SQ" "FETCH_SIZE" "WRITE_SIZE"; do
    echo "== $sc : $c"
    PMC="$c" bash scripts/pmc_quick.sh $sc 2>&1 | grep -v amdgpu.ids | grep "Msamples\|{" | tail -4
  done
done > $OUT/pmc_kernels.txt 2>&1
tail -6 $OUT/pmc_kernels.txt
# 4) BASELINE cfg3 / cfg4 lines, each under the profiler, with the wide parity windows
cd /tmp
for c in cfg3 cfg4; do
  Q=$REPO/gpurun_out/prof_r06z_$c
  rm -rf $Q; mkdir -p $Q
  if [ $c = cfg3 ]; then PAR="--parity-rows 1024 --parity-passes 2"; else PAR="--parity-rows 64 --parity-passes 2"; fi
  echo "python bench.py --config $c $PAR" > $Q/command.txt
  timeout 1500 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config $c $PAR > $Q/trace.log 2>&1
  grep '^{' $Q/trace.log > $OUT/bench_$c.json
  ( cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r06z_$c gpurun_out/r06z/r06z_$c > /dev/null 2>&1 )
done
cd $REPO
# 5) the dispatch sweep on the final dispatcher, the spec kernel's commit histogram (instrumented build), ISA metadata
timeout 900 python scripts/dispatch_sweep.py $OUT/dispatch_sweep.md > $OUT/dispatch_sweep.log 2>&1
tail -4 $OUT/dispatch_sweep.log
[ -f pt-three-ways_amd/libptw_hip_prof.so ] && PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_prof.so timeout 300 python scripts/quick_bench.py cornell,64,64,256,0 cornell,64,64,256,0,seq_small_kernel=3 > $OUT/spec_histogram.txt 2>&1
python - <<'PY'
import json
for name in ("bench_default", "bench_two_ranks_one_gpu", "bench_cfg3", "bench_cfg4", "bench_prefilter_suzanne_perpixel", "bench_prefilter_ce_perpixel",
             "bench_prefilter_cfg4_sequential"):
    try:
        r = json.loads(open(f"gpurun_out/r06z/{name}.json").read().strip().splitlines()[-1])
        keys = ("value", "n_gpus", "rccl_ranks", "value_tile_sharded", "rmse_vs_ref", "samples_word_count_differs", "picks_differ",
                "samples", "parity_rows", "parity_passes", "parity_kernel", "accel_modes")
        print(name, {k: r.get(k) for k in keys if r.get(k) is not None}, "perpixel", (r.get("perpixel_policy") or {}).get("value"),
              "other", [(o.get("config"), o.get("value"), o.get("picks_differ"), (o.get("cpu_baseline") or {}).get("value"), o.get("prefilter_mode")) for o in r.get("other_configs", [])],
              "strict", (r.get("strict_fp") or {}).get("value"), "cpu", (r.get("cpu_baseline") or {}).get("value"),
              "kernel", r["roofline"]["kernel"], "frac", r["roofline"]["frac"], "bytes", len(json.dumps(r)))
    except Exception as e:
        print(name, "FAILED:", e)
PY
