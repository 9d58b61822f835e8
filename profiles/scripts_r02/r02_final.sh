# final validation of the round: every GPU test, smoke(), the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02z; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_cornell1024_full.json 2> $O/bench.err
cat $O/pytest.log $O/smoke.log; head -c 2500 $O/bench_cornell1024_full.json; echo; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02z/bench_cornell1024_full.json'))
print({k:d.get(k) for k in ('value','rmse_vs_ref','max_abs_diff','pixels_bit_identical','samples_word_count_differs','parity_passes','word_count_differences')})
print([(l['cores'],round(l['value'],3)) for l in d['cpu_baseline_legs']], d['perpixel_policy']['value'])
PY
