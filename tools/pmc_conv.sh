#!/bin/bash
# PMC passes for one conv micro-benchmark case.  usage: tools/pmc_conv.sh <shape> <outdir> [only-pattern] [sets: sq|l2|all]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=$1; OUT=$ROOT/$2; ONLY=${3:-"32->32 @128"}; WHICH=${4:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MI355_CONV_SHAPE=$SHAPE
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES"
L2A="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
L2B="FETCH_SIZE"
L2C="WRITE_SIZE"
i=0
for name in SQ1 SQ2 L2A L2B L2C; do
  case $WHICH in sq) [[ $name == SQ* ]] || continue;; l2) [[ $name == L2* ]] || continue;; esac
  set=${!name}
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 $ROOT/tools/bench_kernels.py conv --dtype ${PMC_DTYPE:-bf16} --reps 5 --only "$ONLY" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; }
done
python3 $ROOT/tools/pmc_summary.py $OUT conv_ > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
