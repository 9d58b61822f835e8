timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "perpixel or non_default or soups or row_window or compose or band" 2>&1 | tail -2
timeout 600 python scripts/quick_bench.py suzanne,256,256,64,1 suzanne,256,256,64,1 ce,128,128,16,1 2>&1 | grep -v amdgpu.ids
PTW_PIX_KERNEL=persistent timeout 600 python scripts/quick_bench.py cornell,512,512,256,1 2>&1 | grep -v amdgpu.ids
