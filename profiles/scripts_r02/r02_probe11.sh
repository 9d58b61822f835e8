# PERPIXEL persistent kernel at 4 / 3 / 2 waves per SIMD (432 / 256 / 0 B of scratch per lane), and on Cornell against the lock-step kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in 4 3 2 4 2; do
  echo "== PTW_PIX2_W=$w"; PTW_PIX2_W=$w timeout 300 python scripts/quick_bench.py suzanne,512,512,64,1 ce,256,256,64,1 2>&1 | grep Msamples
  PTW_PIX2_W=$w PTW_PIX_KERNEL=persistent timeout 300 python scripts/quick_bench.py cornell,1024,1024,32,1 2>&1 | grep Msamples
done
echo "== lock-step"; timeout 300 python scripts/quick_bench.py cornell,1024,1024,32,1 2>&1 | grep Msamples
PTW_PIX_KERNEL=legacy timeout 300 python scripts/quick_bench.py suzanne,512,512,64,1 2>&1 | grep Msamples
echo "== bytes"; for w in 4 3 2; do
  PTW_PIX2_W=$w ./pt-three-ways_amd/pt_three_ways_hip -w 32 -h 24 --spp 3 --seed 4 --scene suzanne --rng perpixel --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw
done
