# persistent PERPIXEL kernel: sample pre-claim A/B, then two --pmc passes on Cornell
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in 0 1 0 1; do echo "== OPT=$o"; PTW_PIX2_OPT=$o timeout 300 python scripts/quick_bench.py cornell,1024,1024,16,1 suzanne,384,384,64,1 bbc-owl,768,768,32,1 2>&1 | grep Msamples; done
for o in 0 1; do PTW_PIX2_OPT=$o ./pt-three-ways_amd/pt_three_ways_hip -w 48 -h 32 --spp 5 --seed 4 --scene cornell --rng perpixel --raw --save-every 0 /tmp/g.raw > /dev/null; md5sum < /tmp/g.raw; done
PMC="SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -2
PMC="SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU" bash scripts/pmc_quick.sh cornell,1024,1024,16,1 2>&1 | tail -2
