# persistent PERPIXEL kernel, lean state machine (OPT=2: one surface + scatter + finish block, first-bounce surface rebuilt per sub-sample) vs OPT=0
cd ${GRAFT_REPO_ROOT:-/root/repo}
PTW_PIX2_OPT=2 timeout 600 python -m pytest tests -m gpu -q -k "perpixel or policies or statistical or accel or variants or preview or depth" 2>&1 | tail -3
for o in 0 2 0 2; do for w in 3 4; do echo "== OPT=$o W=$w"; PTW_PIX2_W=$w PTW_PIX2_OPT=$o timeout 300 python scripts/quick_bench.py cornell,1024,1024,16,1 suzanne,384,384,64,1 ce,192,192,32,1 bbc-owl,768,768,32,1 multi-sphere,768,768,32,1 example1,768,768,32,1 2>&1 | grep Msamples; done; done
