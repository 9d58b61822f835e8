# last validation of the round: every GPU test, smoke(), a small bench line through the new traffic keys
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02x; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --width 256 --height 256 --parity-passes 4 > $O/bench_cornell256.json 2> $O/bench.err
cat $O/pytest.log; tail -2 $O/smoke.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02x/bench_cornell256.json'))
print(d['value'], d['roofline']['kernel'], d['roofline']['traffic'], d['rmse_vs_ref'], d['samples_word_count_differs'])
pp=d['perpixel_policy']; print(pp['value'], pp['roofline'])
print([k for k in d.keys()])
PY
tail -3 $O/bench.err
