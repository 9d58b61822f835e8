# quick PMC look at the sequential kernel: bash scripts/pmc_quick.sh "<quick_bench args>"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/pmcq; rm -rf $OUT; mkdir -p $OUT
ARGS=${1:-cornell,128,128,256,0}
COUNTERS=${PMC:-SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA}
rocprofv3 --pmc $COUNTERS -d $OUT -o q -- python $REPO/scripts/quick_bench.py $ARGS > $OUT/log 2>&1
grep Msamples $OUT/log || tail -20 $OUT/log; find $OUT -name "*.db" | head
python3 - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/pmcq/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = list(con.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"))
d = {}
for k, c, v in rows:
    if 'trace' in k:
        d.setdefault(k.split('(')[0][-40:], {})[c] = v
for k, c in d.items():
    print(k, {a: f"{b:.6g}" for a, b in c.items()})
PY
