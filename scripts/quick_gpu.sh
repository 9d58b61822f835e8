timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python scripts/quick_bench.py cornell,512,512,256,1 cornell,512,512,256,1 suzanne,256,256,64,1 ce,128,128,16,1 cornell,128,128,256,0 2>&1 | grep -v amdgpu.ids
