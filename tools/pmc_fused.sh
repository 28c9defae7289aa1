#!/bin/bash
# SQ counter passes over the default bench command (fused engine): instruction mix, issue/wait split.
# Usage (GPU box, repo root): bash tools/pmc_fused.sh <tag>      -> gpurun_out/<tag>/pmc_*/ + <tag>/pmc_fused.json
set -e
TAG=${1:-pmcf}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_IFETCH SQ_INST_CYCLES_SALU" \
         "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_fused_p$i -- python bench.py --steps 3 --warmup 1 --cpu-spp 0 "${@:2}" > $OUT/pmc_fused_p$i.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/pmc_fused.json
echo pmc_done
