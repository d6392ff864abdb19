#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_pmc_march
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SEQ="python3 bench.py --steps 4 --warmup 1 --pipeline 0 --no-cpu-baseline"
timeout -k 10 900 python3 tools/pmc_passes.py $OUT/pmc $OUT/pmc_summary.json \
  --set A=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS \
  --set B=SQ_WAVES,SQ_BUSY_CU_CYCLES,SQ_INSTS_SALU,SQ_ACTIVE_INST_VALU,SQ_THREAD_CYCLES_VALU,SQ_INSTS_SMEM,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS \
  --set C=SQ_WAVES,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_ACTIVE_INST_SCA,GRBM_GUI_ACTIVE \
  --kernel k_field_f16 --kernel k_composite_march --kernel k_march_rays \
  --note "sequential mode, cooperative marcher" -- $SEQ
