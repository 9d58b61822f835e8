# a look at traceSequentialWide on the box: variants side by side + the phase profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
for env in "PTW_WIDE_G=8" "PTW_WIDE_G=16" "PTW_SEQ_WIDE=0" "PTW_WIDE_G=8 PTW_WIDE_CANDIDATES=32" "PTW_WIDE_G=16 PTW_WIDE_CANDIDATES=16"; do
  echo "== $env"; env $env python scripts/quick_bench.py cornell,256,256,256,0 2>&1 | tail -1
done
echo "== phases G=8"; PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_prof.so PTW_WIDE_G=8 python scripts/quick_bench.py cornell,64,64,256,0 2>&1 | grep -E "WIDE|Msamples" | head -10
echo "== phases G=16"; PTW_LIB_PATH=$PWD/pt-three-ways_amd/libptw_hip_prof.so PTW_WIDE_G=16 python scripts/quick_bench.py cornell,64,64,256,0 2>&1 | grep -E "WIDE|Msamples" | head -10
