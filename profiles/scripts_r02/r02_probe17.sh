# persistent PERPIXEL kernel: one shared scatter block + one finish/begin block (OPT=2) against the round-1 state machine (OPT=0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in 0 2 0 2; do for w in 3 4; do echo "== OPT=$o W=$w"; PTW_PIX2_W=$w PTW_PIX2_OPT=$o timeout 300 python scripts/quick_bench.py cornell,1024,1024,16,1 suzanne,384,384,64,1 bbc-owl,768,768,32,1 multi-sphere,768,768,32,1 2>&1 | grep Msamples; done; done
echo "== OPT=2 W=2"; PTW_PIX2_W=2 PTW_PIX2_OPT=2 timeout 300 python scripts/quick_bench.py cornell,1024,1024,16,1 2>&1 | grep Msamples
PTW_PIX2_OPT=2 timeout 600 python -m pytest tests -m gpu -q -k "perpixel or policies or statistical or accel or variants" 2>&1 | tail -3
