# gang kernel (G passes per workgroup in lock step, all eight waves hold primitives) vs two masters
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
for env in "PTW_SEQ_MM=1" "PTW_SEQ_GANG=2" "PTW_SEQ_GANG=4" "PTW_SEQ_GANG=8"; do
  echo "== $env"; env $env timeout 300 python scripts/quick_bench.py suzanne,64,64,2048,0 ce,32,32,2048,0 2>&1 | grep Msamples
done
echo "== bytes"; for env in "PTW_SEQ_MM=0" "PTW_SEQ_GANG=2" "PTW_SEQ_GANG=4" "PTW_SEQ_GANG=8"; do
  env $env ./pt-three-ways_amd/pt_three_ways_hip -w 24 -h 18 --spp 11 --seed 4 --scene suzanne --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw
  env $env PTW_STAGE_BUDGET_KB=8 ./pt-three-ways_amd/pt_three_ways_hip -w 16 -h 8 --spp 5 --seed 4 --scene ce --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw
done
echo "== phases ce"; for env in "PTW_SEQ_MM=0" "PTW_SEQ_MM=1"; do env $env PTW_LIB_PATH=$L/libptw_hip_prof.so timeout 120 python scripts/quick_bench.py ce,32,32,256,0 2>&1 | grep -E "PHASES|WORKER|Msamples" | head -6; done
