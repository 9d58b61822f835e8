# full-size SEQUENTIAL lines of the configurations beyond 128 triangles with the two-master kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02p; mkdir -p $O
python bench.py --scene suzanne --spp 512 --no-parity > $O/bench_suzanne1024_512spp_full.json 2> $O/bench.err
python bench.py --scene ce --width 2048 --height 2048 --spp 1024 --rows 0:128 --no-parity --no-cpu-baseline --no-secondary > $O/bench_ce2048_1024spp_sequential_rows128.json 2>> $O/bench.err
for a in suzanne,128,128,512,0 ce,32,32,1024,0; do for c in FETCH_SIZE WRITE_SIZE; do
  echo "== $a $c"; PMC=$c bash scripts/pmc_quick.sh $a 2>&1 | tail -3; done; done > $O/pmc.log 2>&1
head -c 900 $O/bench_suzanne1024_512spp_full.json; echo; head -c 900 $O/bench_ce2048_1024spp_sequential_rows128.json; echo; cat $O/pmc.log; tail -5 $O/bench.err
