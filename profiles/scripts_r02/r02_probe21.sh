# persistent PERPIXEL kernel with the fan-out state in LDS (default build) vs in registers (alt build = previous commit)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
S="cornell,1024,1024,16,1 suzanne,384,384,64,1 ce,192,192,32,1 bbc-owl,768,768,32,1 multi-sphere,768,768,32,1 example1,768,768,32,1"
for lib in libptw_hip.so libptw_hip_alt.so libptw_hip.so libptw_hip_alt.so; do echo "== $lib"; PTW_LIB_PATH=$L/$lib timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples; done
echo "== W=3"; PTW_PIX2_W=3 timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples
timeout 600 python -m pytest tests -m gpu -q -k "perpixel or policies or statistical or variants or preview or depth" 2>&1 | tail -3
for c in FETCH_SIZE WRITE_SIZE; do echo "== cornell,1024,1024,16,1 $c"; PMC=$c bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -1; done
