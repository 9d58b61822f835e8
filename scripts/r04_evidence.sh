#!/bin/bash
# RETIRED (round 6): this script drives the library through the PTW_SEQ_* / PTW_PIX_* / PTW_TEST_*
# environment switches of rounds 2-4.  ABI v5 (commit 916a1dc is the last with them) replaced those by
# ptw_debug_options (`--debug name=value` in the CLI and bench.py, Context.set_debug in Python): run against
# HEAD it would time the DEFAULT dispatch under the old labels.  Kept as the record of how profiles/r04* were
# taken; to re-run it, check out 916a1dc.
if [ "${PTW_ALLOW_RETIRED_SCRIPT:-0}" != "1" ]; then
  echo "$0: retired - needs commit 916a1dc (the PTW_SEQ_*/PTW_PIX_* environment switches are gone; use --debug)" >&2
  exit 2
fi
# Round 4 evidence run (one gpurun call) on the library AS SHIPPED: the whole GPU suite, the default
# bench line, rocprofv3 --kernel-trace --stats of the same command, the PMC passes behind
# profiles/hbm_traffic.json (sequential + the lock-step PERPIXEL kernel), BASELINE cfg3 / cfg4 under the
# profiler - cfg3 with ONE WHOLE-FRAME parity comparison (rows [0, 1024) x 2 passes), cfg4 with rows
# [0, 64) x 2 passes - and bench.py --gpus 2 with two ranks on the one GPU.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04z
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
[ -n "$SKIP_PYTEST" ] || ( timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log )
tail -4 $OUT/pytest_gpu.log
# 1) the default line, as the driver runs it at N = 1 (steps 1 here)
( timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
tail -c 300 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
# 1b) N = 2 on this one GPU: bench.py launches its two ranks itself
( PTW_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --width 512 --height 512 --no-cpu-baseline --no-parity > $OUT/bench_two_ranks_one_gpu.json 2> $OUT/bench_two_ranks_one_gpu.err; echo "rc=$?" >> $OUT/bench_two_ranks_one_gpu.err )
tail -c 300 $OUT/bench_two_ranks_one_gpu.json; tail -2 $OUT/bench_two_ranks_one_gpu.err
# 2) the same command under rocprofv3 (CPU legs and the child-process leg left out: they launch no kernels of this process)
cd /tmp && export TMPDIR=/tmp
P=$REPO/gpurun_out/prof_r04z
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
cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r04z gpurun_out/r04z/r04z_default > /dev/null 2>&1
# 3b) the lock-step PERPIXEL kernel alone at the BASELINE frame (64 of its 256 passes): HBM bytes per sample
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== tracePerPixel (lock step) cornell,1024,1024,64,1 $c"
  PTW_PIX_KERNEL=legacy PMC="$c" bash scripts/pmc_quick.sh cornell,1024,1024,64,1 2>&1 | grep -v amdgpu.ids | grep "Msamples\|PerPixel"
done > $OUT/pmc_lockstep_perpixel.txt 2>&1
cat $OUT/pmc_lockstep_perpixel.txt
# 4) BASELINE cfg3 / cfg4 lines, each under the profiler, with the wide parity windows
cd /tmp
for c in cfg3 cfg4; do
  Q=$REPO/gpurun_out/prof_r04z_$c
  rm -rf $Q; mkdir -p $Q
  if [ $c = cfg3 ]; then PAR="--parity-rows 1024 --parity-passes 2"; else PAR="--parity-rows 64 --parity-passes 2"; fi
  echo "python bench.py --config $c --no-cpu-baseline $PAR" > $Q/command.txt
  timeout 1500 rocprofv3 --kernel-trace --stats -d $Q/trace -o trace -- python $REPO/bench.py --config $c --no-cpu-baseline $PAR > $Q/trace.log 2>&1
  grep '^{' $Q/trace.log > $OUT/bench_$c.json
  ( cd $REPO && python scripts/summarize_prof.py gpurun_out/prof_r04z_$c gpurun_out/r04z/r04z_$c > /dev/null 2>&1 )
done
cd $REPO
ls -la $OUT
python - <<'PY'
import json
for name in ("bench_default", "bench_two_ranks_one_gpu", "bench_cfg3", "bench_cfg4"):
    try:
        r = json.loads(open(f"gpurun_out/r04z/{name}.json").read().strip().splitlines()[-1])
        keys = ("value", "n_gpus", "rccl_ranks", "value_tile_sharded", "rmse_vs_ref", "samples_word_count_differs", "samples", "parity_rows", "parity_passes", "parity_kernel")
        print(name, {k: r.get(k) for k in keys}, "perpixel", (r.get("perpixel_policy") or {}).get("value"),
              "other", [(o.get("config"), o.get("value")) for o in r.get("other_configs", [])], "strict", (r.get("strict_fp") or {}).get("value"),
              "bytes", len(json.dumps(r)))
    except Exception as e:
        print(name, "FAILED:", e)
PY
