# round-2 evidence run on the box: tests, the default bench, rocprofv3 of the same command, side configs
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
python bench.py > $O/bench_cornell1024_full.json 2> $O/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_full -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-secondary > $GRAFT_REPO_ROOT/$O/trace_full.log 2>&1 )
python bench.py --scene suzanne --spp 512 --no-parity > $O/bench_suzanne1024_512spp_full.json 2>> $O/bench.err
python bench.py --scene ce --width 2048 --height 2048 --spp 1024 --policy perpixel --rows 0:128 --no-parity --no-cpu-baseline > $O/bench_ce2048_1024spp_perpixel_rows128.json 2>> $O/bench.err
bash scripts/profile_gpu.sh r02e > $O/profile.log 2>&1
for s in cornell single-sphere multi-sphere example1; do for e in "PTW_SEQ_WIDE=0" "PTW_SEQ_WIDE=1"; do echo "== $s $e"; env $e python scripts/quick_bench.py $s,128,128,256,0 2>&1 | tail -1; done; done > $O/wide_vs_spec_scenes.log 2>&1
cat $O/pytest.log; head -c 1500 $O/bench_cornell1024_full.json; echo; head -c 600 $O/bench_suzanne1024_512spp_full.json; echo; head -c 600 $O/bench_ce2048_1024spp_perpixel_rows128.json; echo; cat $O/wide_vs_spec_scenes.log; find $O/trace_full -name "*kernel_stats.csv" | head -2 | xargs -I{} head -6 {}
