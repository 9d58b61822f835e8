mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for lib in libptw_hip.so libptw_hip_single.so; do echo "== $lib"; PTW_LIB_PATH=$PWD/pt-three-ways_amd/$lib timeout 600 python scripts/quick_bench.py cornell,128,128,256,0 suzanne,64,64,256,0 ce,16,16,256,0 2>&1 | grep -v amdgpu.ids; done
