#!/bin/bash
# rocprofv3 PMC passes over the ABA f32 kernel (MIT humanoid, B=262144); run on the GPU box from the repo root.
# usage: tools/pmc_run.sh <outdir under gpurun_out> [kernel: aba|rnea] [precision: 32|64]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc}
KIND=${2:-aba}
PREC=${3:-32}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
SETS_FROM=${PMC_FROM:-1}
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64" \
           "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQC_TC_STALL"; do
  i=$((i+1))
  if [ $i -lt $SETS_FROM ] || [ $i -gt ${PMC_TO:-99} ]; then continue; fi
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/pmc_target.py $KIND $PREC ${PMC_MODEL:-mit_humanoid} > $OUT/p$i.log 2>&1
done
python3 $ROOT/tools/pmc_summarize.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
