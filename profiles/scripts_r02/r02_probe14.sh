# persistent PERPIXEL kernel: triangle-loop forms (0 = rotate, 1 = two per trip, 2 = select, 3 = both) x waves per SIMD
cd ${GRAFT_REPO_ROOT:-/root/repo}
for l in 0 1 2 3 0; do for w in 3 4; do
  echo "== LOOP=$l W=$w"; PTW_PIX2_LOOP=$l PTW_PIX2_W=$w timeout 300 python scripts/quick_bench.py cornell,1024,1024,16,1 suzanne,384,384,64,1 example1,768,768,32,1 2>&1 | grep Msamples
done; done
echo "== bytes"; for l in 0 1 2 3; do
  PTW_PIX2_LOOP=$l ./pt-three-ways_amd/pt_three_ways_hip -w 32 -h 24 --spp 3 --seed 4 --scene suzanne --rng perpixel --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw
  PTW_PIX2_LOOP=$l ./pt-three-ways_amd/pt_three_ways_hip -w 48 -h 32 --spp 5 --seed 4 --scene cornell --rng perpixel --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw
done
