# instruction counts of the persistent PERPIXEL kernel on Cornell (one --pmc pass) + the new variant tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_round2.py -m gpu -q -k "variants or two_master or variant_is_reported or statistical" 2>&1 | tail -4
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -3
PMC="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_FLAT" bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -3
PMC="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" PTW_PIX_KERNEL=legacy bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -3
