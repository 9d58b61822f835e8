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
# round 4, second GPU call: the decoupled two-master protocol - parity, then A/B against the lock step
# (libptw_hip_alt.so = the same tree with -DPTW_SEQ_DECOUPLED=0) on cfg3 / cfg4; the headline and the
# PERPIXEL kernels after the scalar-constant change; the RCCL peer-exit test.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_cli.py -x -q -m gpu -k "two_master or raw or kernel_variants or sequential" > $O/pytest_mm.log 2>&1; echo "pytest_mm rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_mm.log
timeout 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "rccl_peer" > $O/pytest_peer.log 2>&1; echo "pytest_peer rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_peer.log
run() { # name, lib, args...
  local name=$1 lib=$2; shift 2
  PTW_LIB_PATH=$PWD/pt-three-ways_amd/$lib timeout 600 python bench.py "$@" --no-cpu-baseline --no-parity > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY' 2>&1 | tee -a gpurun_out/r04b/summary.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.3f" % r["value"], r["roofline"]["kernel"], "frac %.4f" % r["roofline"]["frac"], "ms/launch %.1f" % r["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run cfg3_decoupled libptw_hip.so --config cfg3
run cfg3_lockstep libptw_hip_alt.so --config cfg3
run cfg4_decoupled libptw_hip.so --config cfg4
run cfg4_lockstep libptw_hip_alt.so --config cfg4
run cfg3_decoupled_2 libptw_hip.so --config cfg3
run headline libptw_hip.so --no-secondary --no-other-configs --no-strict
run perpixel libptw_hip.so --policy perpixel --steps 2 --warmup 1
run headline_strict libptw_hip_strict.so --no-secondary --no-other-configs --no-strict
