# worker-wave kernels with the look-ahead scatter (the master evaluates the next sub-sample's first-bounce scatter while the workers search)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity.py -m gpu -q -x -k "two_master or suzanne or ce or soup or golden or scene or depth or fan" 2>&1 | tail -3
timeout 300 python scripts/quick_bench.py suzanne,128,128,512,0 suzanne,128,128,256,0 ce,48,48,1024,0 ce,64,64,256,0 2>&1 | grep Msamples
echo "== phases"; PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 120 python scripts/quick_bench.py suzanne,64,64,512,0 2>&1 | grep -E "PHASES|WORKER|Msamples" | head -4
