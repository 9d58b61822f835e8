timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/quick_bench.py cornell,128,128,256,0 cornell,128,128,256,0 suzanne,64,64,256,0 ce,16,16,256,0 cornell,256,256,256,1 2>&1 | grep -v amdgpu.ids
