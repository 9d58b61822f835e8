# speculative kernel with kill flags (default build) vs without (alt build = previous commit)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
S="cornell,256,256,256,0 single-sphere,128,128,256,0 multi-sphere,128,128,256,0 example1,128,128,256,0 bbc-owl,128,128,256,0"
for lib in libptw_hip.so libptw_hip_alt.so libptw_hip.so libptw_hip_alt.so; do echo "== $lib"; PTW_LIB_PATH=$L/$lib timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -q -x 2>&1 | tail -3
echo "== phases"; PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 120 python scripts/quick_bench.py cornell,128,128,256,0 2>&1 | grep -E "SPEC wave|Msamples" | head -10
