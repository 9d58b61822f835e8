# the eight-wave speculative kernel vs the four-wave one
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
timeout 300 python -m pytest tests/test_gpu_cli.py -m gpu -q -x -k "variants" 2>&1 | tail -4
for env in "PTW_SEQ_SPEC8=1" "PTW_SEQ_SPEC8=0"; do
  echo "== $env"; env $env timeout 120 python scripts/quick_bench.py cornell,256,256,256,0 single-sphere,128,128,256,0 example1,128,128,256,0 2>&1 | grep Msamples
done
echo "== phases"; PTW_SEQ_SPEC8=1 PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 120 python scripts/quick_bench.py cornell,128,128,256,0 2>&1 | grep -E "SPEC8|Msamples" | head -12
