# persistent PERPIXEL kernel, small scenes: generator + sample ids in LDS too (128 VGPRs at 4 waves per SIMD, nothing spilled)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -m pytest tests -m gpu -q -k "perpixel or policies or statistical or variants or preview or interleaved" 2>&1 | tail -2
timeout 200 python scripts/quick_bench.py cornell,1024,1024,16,1 bbc-owl,768,768,32,1 multi-sphere,768,768,32,1 example1,768,768,32,1 single-sphere,768,768,32,1 2>&1 | grep Msamples
for c in FETCH_SIZE WRITE_SIZE; do echo "== cornell,1024,1024,16,1 $c"; PMC=$c bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -1; done
