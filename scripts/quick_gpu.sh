mkdir -p gpurun_out
timeout 1500 python bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_full_r01e.json
timeout 900 python bench.py --scene suzanne --spp 512 --width 256 --height 256 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_suzanne_r01e.json
timeout 900 python bench.py --scene ce --spp 1024 --width 64 --height 64 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_ce_r01e.json
bash scripts/profile_gpu.sh r01e > gpurun_out/profile_r01e.log 2>&1
