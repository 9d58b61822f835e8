# PERPIXEL lock-step kernel: triangles through scalar loads (SGPR operands), 4 / 3 / 2 waves per SIMD
cd ${GRAFT_REPO_ROOT:-/root/repo}
for env in "PTW_PIX_SLOAD=0" "PTW_PIX_SLOAD=1" "PTW_PIX_SLOAD=3" "PTW_PIX_SLOAD=2" "PTW_PIX_SLOAD=0" "PTW_PIX_SLOAD=3"; do
  echo "== $env"; env $env timeout 300 python scripts/quick_bench.py cornell,1024,1024,32,1 cornell,1024,1024,32,1 example1,512,512,32,1 2>&1 | grep Msamples
done
echo "== bytes"; for env in "PTW_PIX_SLOAD=0" "PTW_PIX_SLOAD=1" "PTW_PIX_SLOAD=3" "PTW_PIX_SLOAD=2"; do
  env $env ./pt-three-ways_amd/pt_three_ways_hip -w 64 -h 48 --spp 7 --seed 4 --scene cornell --rng perpixel --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw
  env $env PTW_PIX_KERNEL=legacy ./pt-three-ways_amd/pt_three_ways_hip -w 32 -h 24 --spp 3 --seed 4 --scene suzanne --rng perpixel --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw
done
