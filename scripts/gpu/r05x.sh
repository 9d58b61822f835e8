#!/bin/bash
# round 5, closing check on the FINAL tree: smoke(), pytest -m gpu, and the two-rank line with RCCL's channel
# lines in it (bench.py's log routing became a function after the r05w call)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${1:-r05x}; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
( PTW_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --width 512 --height 512 --no-cpu-baseline --no-parity > $O/bench_two_ranks_one_gpu.json 2> $O/two.err; echo "two ranks rc=$?" )
python - <<PY
import json
line=[l for l in open("$O/bench_two_ranks_one_gpu.json") if l.startswith("{")][-1]
d=json.loads(line); print(d["value"], d.get("value_tile_sharded"), d["rccl_transport"], len(line))
PY
