timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edge or many_passes or argument" 2>&1 | tail -12
