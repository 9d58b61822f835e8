# after the look-ahead scatter / slot-major mapping: every GPU test, smoke(), timing of the worker-wave kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02v; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 300 python scripts/quick_bench.py suzanne,128,128,512,0 suzanne,128,128,256,0 ce,48,48,1024,0 ce,64,64,256,0 cornell,256,256,256,0 2>&1 | grep Msamples > $O/timing.log
cat $O/pytest.log; tail -1 $O/smoke.log; cat $O/timing.log
