# wide v2b (setprio, lean commit, symmetric sincos) vs spec; perpixel occupancy A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
for env in "PTW_WIDE_G=8" "PTW_WIDE_G=16" "PTW_SEQ_WIDE=0"; do
  echo "== $env"; env $env python scripts/quick_bench.py cornell,256,256,256,0 2>&1 | tail -1
done
echo "== phases G=8"; PTW_LIB_PATH=$L/libptw_hip_prof.so PTW_WIDE_G=8 python scripts/quick_bench.py cornell,64,64,256,0 2>&1 | grep -E "WIDE|Msamples" | head -10
echo "== spec phases"; PTW_LIB_PATH=$L/libptw_hip_prof.so PTW_SEQ_WIDE=0 python scripts/quick_bench.py cornell,64,64,256,0 2>&1 | grep -E "SPEC|Msamples" | head -10
for lib in libptw_hip.so libptw_hip_pw33.so libptw_hip_pw32.so; do
  echo "== perpixel $lib"; PTW_LIB_PATH=$L/$lib python scripts/quick_bench.py cornell,512,512,256,1 suzanne,256,256,128,1 ce,64,64,256,1 2>&1 | grep Msamples
done
echo "== perpixel persistent on cornell"; PTW_PIX_KERNEL=persistent PTW_LIB_PATH=$L/libptw_hip_pw32.so python scripts/quick_bench.py cornell,512,512,256,1 2>&1 | grep Msamples
echo "== seq suzanne/ce"; python scripts/quick_bench.py suzanne,128,128,512,0 ce,32,32,1024,0 2>&1 | grep Msamples
