#!/bin/bash
# round 6 (third session): the unit-level u-first early-out as shipped (run-time flag from the host-side statistic,
# ptw_debug_options.seq_unit_ufirst forces it): the whole GPU suite; ce / suzanne with the flag at its
# default, forced off and forced on; cfg4's and cfg3's bench lines.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06ab; mkdir -p $out
( timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log )
grep -E "passed|failed|rc=" $out/pytest_gpu.log | tail -3
for rep in 1 2; do
for sc in ce,2048,8,1024,0 ce,2048,4,256,0 suzanne,1024,128,512,0 suzanne,1024,64,256,0; do
  python scripts/quick_bench.py $sc $sc,seq_unit_ufirst=0 $sc,seq_unit_ufirst=1 2>&1 | grep "Msamples\|rror"
done
done | tee $out/unit_ufirst_flag_ab.txt
( timeout 900 python bench.py --config cfg4 --no-cpu-baseline --parity-passes 2 > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "rc=$?" >> $out/bench_cfg4.err )
( timeout 900 python bench.py --config cfg3 --no-cpu-baseline --parity-passes 2 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "rc=$?" >> $out/bench_cfg3.err )
python - <<'PY'
import json
for c in ("cfg4", "cfg3"):
    try:
        r = json.loads(open(f"gpurun_out/r06ab/bench_{c}.json").read().strip().splitlines()[-1])
        print(c, r["value"], r["roofline"]["kernel"], r["roofline"]["frac"], r.get("samples_word_count_differs"), r.get("picks_differ"), r.get("samples"))
    except Exception as e:
        print(c, "FAILED", e)
PY
