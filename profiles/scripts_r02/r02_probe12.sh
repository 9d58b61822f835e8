# PERPIXEL: lock-step vs persistent (3 / 4 waves per SIMD) on every small scene
cd ${GRAFT_REPO_ROOT:-/root/repo}
S="cornell,768,768,32,1 single-sphere,768,768,32,1 multi-sphere,768,768,32,1 example1,768,768,32,1 bbc-owl,768,768,32,1"
echo "== lock-step"; PTW_PIX_KERNEL=legacy timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples
echo "== persistent 3"; PTW_PIX2_W=3 PTW_PIX_KERNEL=persistent timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples
echo "== persistent 4"; PTW_PIX2_W=4 PTW_PIX_KERNEL=persistent timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples
echo "== lock-step"; PTW_PIX_KERNEL=legacy timeout 300 python scripts/quick_bench.py $S 2>&1 | grep Msamples
