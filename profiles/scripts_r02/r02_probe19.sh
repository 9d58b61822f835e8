# pickNearest: scalar walk over up to 6 candidates (default build) vs the wave-wide reductions from 3 candidates on (alt build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
S="cornell,256,256,256,0 suzanne,128,128,512,0 ce,48,48,1024,0 example1,128,128,256,0 multi-sphere,128,128,256,0"
for lib in libptw_hip.so libptw_hip_alt.so libptw_hip.so libptw_hip_alt.so; do echo "== $lib"; PTW_LIB_PATH=$L/$lib timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
echo "== phases ce"; PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 120 python scripts/quick_bench.py ce,32,32,512,0 2>&1 | grep -E "PHASES|WORKER|Msamples" | head -4
