#!/bin/bash
# round 6, closing call: smoke() and the plan / ABI tests on the final library; A/B of a second ballot after v in the
# unit-level early-out (variant build -DPTW_SEQ_UNIT_V_BALLOT=1, not shipped) on ce and the 24 202-triangle mesh.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06ag; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $out/smoke.txt
( timeout 600 python -m pytest tests -q -m gpu -k "unit_early_out or baseline_scenes or abi or dispatch" > $out/pytest_subset.log 2>&1; echo "rc=$?" >> $out/pytest_subset.log ); grep -E "passed|failed|rc=" $out/pytest_subset.log | tail -2
for rep in 1 2; do
  for lib in libptw_hip.so libptw_hip_pwvb.so; do
    echo "== $lib"
    PTW_LIB_PATH=$PWD/pt-three-ways_amd/$lib python scripts/quick_bench.py ce,2048,8,1024,0 ce,2048,4,256,0 2>&1 | grep "Msamples\|rror"
  done
done | tee $out/unit_v_ballot_ab.txt
for lib in libptw_hip.so libptw_hip_pwvb.so; do
  echo "== $lib"
  PTW_LIB_PATH=$PWD/pt-three-ways_amd/$lib python scripts/big_mesh_bench.py 5 2>&1 | grep "sequential.*rule"
done | tee -a $out/unit_v_ballot_ab.txt
