# final validation of the round (after the two-master and PERPIXEL work): every GPU test, smoke(), the default bench
# line, the PERPIXEL kernel's HBM traffic (two --pmc passes), the suzanne line again (its PERPIXEL half changed)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02y; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_cornell1024_full.json 2> $O/bench.err
for c in FETCH_SIZE WRITE_SIZE; do echo "== cornell,1024,1024,16,1 $c"; PMC=$c bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -3; done > $O/pmc_perpixel.log 2>&1
cat $O/pytest.log $O/smoke.log $O/pmc_perpixel.log; head -c 1200 $O/bench_cornell1024_full.json; echo; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02y/bench_cornell1024_full.json'))
print({k:d.get(k) for k in ('value','rmse_vs_ref','max_abs_diff','pixels_bit_identical','samples_word_count_differs','parity_passes','word_count_differences')})
print([(l['cores'],round(l['value'],3)) for l in d['cpu_baseline_legs']], d['perpixel_policy']['value'], d['perpixel_policy']['roofline'])
PY
