timeout 600 python scripts/quick_bench.py cornell,128,128,256,0 cornell,64,64,1024,0 cornell,128,128,256,0 cornell,64,64,1024,0 cornell,128,128,256,0 cornell,64,64,1024,0 2>&1 | grep -v amdgpu.ids
