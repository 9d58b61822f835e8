mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4; done
timeout 600 python scripts/quick_bench.py cornell,128,128,256,0 cornell,256,256,256,1 suzanne,64,64,256,0 suzanne,128,128,64,1 ce,16,16,256,0 ce,64,64,16,1 2>&1 | grep -v amdgpu.ids
