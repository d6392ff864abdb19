#!/bin/bash
# Round-3 first probe: GPU tests, reference bench lines (driver's command, frame groups on one GPU), counter passes over the marchers and the field kernel.
set -o pipefail
OUT=gpurun_out/r03_probe1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err && echo "bench driver cmd done" &&
for cfg in "--steps 384" "--steps 384 --group-frames 2" "--steps 384 --group-frames 4" "--steps 20 --pipeline 0"; do
  timeout -k 10 200 python3 bench.py $cfg --no-cpu-baseline > "$OUT/bench_$(echo $cfg | tr -d ' -').json" 2>/dev/null || exit 1
  echo "bench $cfg done"
done
rocprofv3 -L > $OUT/counters_all.txt 2>&1
SEQ="python3 bench.py --steps 4 --warmup 1 --pipeline 0 --no-cpu-baseline"
timeout -k 10 900 python3 tools/pmc_passes.py $OUT/pmc_seq $OUT/pmc_seq_summary.json \
  --set A=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS \
  --set B=SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_INSTS_MFMA,SQ_ACTIVE_INST_VALU,SQ_THREAD_CYCLES_VALU,SQ_INSTS_SALU,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS \
  --set C=SQ_WAVES,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_VALU_MFMA_COEXEC_CYCLES,SQ_ACTIVE_INST_SCA \
  --set D=GRBM_GUI_ACTIVE,GRBM_COUNT \
  --kernel k_field_f16 --kernel k_composite_march --kernel k_march_rays --kernel k_composite_rays --kernel k_scatter_advance --kernel k_loop \
  --note "sequential mode (one frame at a time), 4 timed frames + warm-up + count of the 20-frame test sequence" -- $SEQ
echo "pmc rc $?"
ls $OUT
