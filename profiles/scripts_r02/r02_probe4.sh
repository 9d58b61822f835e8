# (1) which numerical shortcut causes the one decision flip of the full frame; (2) the accelerated mode
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/pt-three-ways_amd
for lib in libptw_hip.so libptw_hip_ieee.so libptw_hip_nofma.so libptw_hip_strict.so; do
  PTW_LIB_PATH=$L/$lib timeout 200 python scripts/flip_probe.py 2>&1 | tail -1
done
timeout 300 python -m pytest tests/test_gpu_accel.py -m gpu -q 2>&1 | tail -5
O=gpurun_out/r02g; mkdir -p $O
timeout 300 python bench.py --scene suzanne --spp 512 --policy perpixel --accel bvh --no-parity --no-cpu-baseline > $O/bench_suzanne1024_512spp_perpixel_bvh.json 2> $O/err.log
timeout 600 python bench.py --scene ce --width 2048 --height 2048 --spp 1024 --policy perpixel --accel bvh --no-parity --no-cpu-baseline > $O/bench_ce2048_1024spp_perpixel_bvh_full.json 2>> $O/err.log
timeout 300 python bench.py --policy perpixel --accel bvh --no-parity --no-cpu-baseline > $O/bench_cornell1024_perpixel_bvh.json 2>> $O/err.log
for f in $O/*.json; do echo $f; head -c 700 $f; echo; done; tail -5 $O/err.log
