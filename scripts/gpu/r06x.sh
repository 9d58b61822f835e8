#!/bin/bash
# round 6 (second session): the shares by place from 12 units on (inside the equal shares' instantiation below 31):
# the whole GPU suite, cfg3's line, the sweep of the unit counts in between against equal shares.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06x; mkdir -p $out
( timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log )
grep -E "passed|failed|rc=" $out/pytest_gpu.log | tail -3
python scripts/quick_bench.py suzanne,1024,128,512,0 suzanne,1024,128,512,0,seq_units=3:3:3 ce,2048,8,1024,0 2>&1 | grep Msamples | tee $out/suzanne_ce.txt
( timeout 900 python bench.py --config cfg3 --no-cpu-baseline --parity-passes 2 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "rc=$?" >> $out/bench_cfg3.err )
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06x/bench_cfg3.json").read().strip().splitlines()[-1])
print("cfg3", r["value"], r["roofline"]["kernel"], r["roofline"]["frac"], r.get("samples_word_count_differs"), r.get("picks_differ"), r.get("samples"))
PY
SWEEP_SIZES=800,900,1000,1100,1300,1400,1600,1750,1900 SWEEP_PASSES=512 timeout 900 python scripts/dispatch_sweep.py $out/sweep_shares_by_place_from_12_units.md > $out/sweep.log 2>&1
grep "sequential" $out/sweep_shares_by_place_from_12_units.md | grep -v "one master" | cut -d'|' -f3,4,6,7,9,10,11,12
