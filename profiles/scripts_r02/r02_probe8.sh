# two masters per workgroup (PTW_SEQ_MM=1) vs one, scenes beyond 128 triangles
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -q -x -k "two_master" 2>&1 | tail -4
for env in "PTW_SEQ_MM=0" "PTW_SEQ_MM=1"; do
  echo "== $env"; env $env timeout 300 python scripts/quick_bench.py suzanne,128,128,512,0 suzanne,128,128,256,0 ce,48,48,1024,0 ce,64,64,256,0 2>&1 | grep Msamples
done
echo "== phases"; for env in "PTW_SEQ_MM=0" "PTW_SEQ_MM=1"; do env $env PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 120 python scripts/quick_bench.py suzanne,64,64,256,0 2>&1 | grep -E "PHASES|WORKER|Msamples" | head -6; done
