cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
for lib in libptw_hip.so libptw_hip_ss.so; do
  echo "== $lib"; PTW_LIB_PATH=$L/$lib timeout 200 python scripts/quick_bench.py suzanne,128,128,512,0 ce,32,32,1024,0 example1,128,128,512,0 2>&1 | grep Msamples
done
PTW_LIB_PATH=$L/libptw_hip_ss.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sequential or soup or non_default" 2>&1 | tail -2
