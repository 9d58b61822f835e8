timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python scripts/quick_bench.py suzanne,64,64,256,0 suzanne,64,64,512,0 ce,16,16,256,0 ce,16,16,1024,0 cornell,128,128,256,0 2>&1 | grep -v amdgpu.ids
