mkdir -p gpurun_out
timeout 900 python bench.py --width 256 --height 256 --steps 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_small.json
