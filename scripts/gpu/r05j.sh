#!/bin/bash
# round 5: smoke of the bench.py paths the final evidence run relies on (side configurations with picks and
# CPU legs at reduced sizes, two ranks on one GPU with rccl_transport) and of the new GPU tests
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${1:-r05j}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu -k "bench_gpus_2 or missing_rank or describes or cfg1 or paired" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import sys, json
sys.argv = ["bench.py"]
import bench, torch
bench.CONFIGS["cfg3"].update(width=128, height=128, spp=300)
bench.CONFIGS["cfg4"].update(width=64, height=64, spp=300, rows="0:8")
bench.SIDE_PARITY["cfg3"] = dict(rows_end=16, passes=3)
bench.SIDE_PARITY["cfg4"] = dict(rows_end=4, passes=3)
bench.CPU_SAMPLE_FRAME.update(suzanne=32, ce=8)
pkg = bench.entry.load_package()
sys.path.insert(0, str(bench.ROOT / "tests"))
import oracle_binding as ob
for name in ("cfg3", "cfg4"):
    r = bench.side_config(pkg, ob, name, 0, 6, True, True, 6)
    print(json.dumps({k: r[k] for k in ("config", "value", "kernel", "samples_word_count_differs", "picks_differ", "parity_kernel", "cpu_baseline", "vs_cpu_6t")}))
PY
