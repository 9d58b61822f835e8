# PERPIXEL A/B on Cornell: branch-free tests, occupancy
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
for lib in libptw_hip.so libptw_hip_sel.so libptw_hip_sel3.so libptw_hip_w5.so; do
  echo "== $lib"; PTW_LIB_PATH=$L/$lib timeout 100 python scripts/quick_bench.py cornell,1024,1024,64,1 example1,512,512,64,1 2>&1 | grep Msamples
done
PTW_LIB_PATH=$L/libptw_hip_sel.so timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "perpixel" 2>&1 | tail -2
