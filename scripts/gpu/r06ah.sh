#!/bin/bash
# round 6, last kernel change: the unit-level early-out with a second ballot after v in the resident slots (shipped
# after r06ag's A/B).  Everything that runs the <...,unit> instantiations against the oracle - all 26 instantiations
# forced on, the exact-tie scenes, ce and suzanne on / off / by the rule, cfg4's full-width prefix with picks, the
# 24 202-triangle OBJ scene - and cfg4's bench line with the wide parity window.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06ah; mkdir -p $out
( timeout 420 python -m pytest tests -q -m gpu -k "unit or ties or baseline_scenes or cfg4_full or obj_scene or dropped" --durations=4 > $out/pytest_unit_kernels.log 2>&1; echo "rc=$?" >> $out/pytest_unit_kernels.log )
grep -E "passed|failed|rc=" $out/pytest_unit_kernels.log | tail -3
( timeout 300 python bench.py --config cfg4 --no-cpu-baseline --parity-rows 64 --parity-passes 2 > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "rc=$?" >> $out/bench_cfg4.err )
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06ah/bench_cfg4.json").read().strip().splitlines()[-1])
print("cfg4", r["value"], r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r.get("samples_word_count_differs"), r.get("picks_differ"), r.get("samples"), r.get("parity_rows"))
PY
