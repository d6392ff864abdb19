#!/bin/bash
# Collects round 4's measurement artefacts on the GPU box into gpurun_out/profiles_r04/ (the summaries are copied into profiles/
# afterwards).  rocprofv3: the profiled program comes right after `--`; counters in their own passes (with --kernel-trace only).
#   bash tools/collect_profiles_r04.sh [bench|stats|pmc|grid|fp32|pmc32|cpu800|modes|all]
set -o pipefail
OUT=gpurun_out/profiles_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PART=${1:-all}
want() { [ "$PART" = "all" ] || [ "$PART" = "$1" ]; }
stats() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -o p -- python3 bench.py "$@" > $OUT/bench_${name}_under_rocprof.json 2> $OUT/${name}.err || return 1
  cp $(find $OUT/tmp_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/tmp_$name
}
if want bench; then
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>$OUT/bench_driver_cmd.err || exit 1; echo "driver cmd done"
python3 bench.py > $OUT/bench_default.json 2>/dev/null || exit 1; echo "bench default done"
python3 bench.py --steps 20 --pipeline 0 --group-frames 1 --no-cpu-baseline > $OUT/bench_sequential.json 2>/dev/null || exit 1
python3 bench.py --steps 20 --warmup 5 --pipeline 0 --no-cpu-baseline --no-secondary > $OUT/bench_one_loop_at_a_time.json 2>/dev/null || exit 1
python3 bench.py --fp32 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_fp32.json 2>/dev/null || exit 1
echo "bench lines done"
fi
if want stats; then
stats default --steps 20 --warmup 5 --no-cpu-baseline --no-secondary || exit 1
# the F = 4 ONE-LOOP-AT-A-TIME command: the headline's exclusive fraction reproducible from a rocprof CSV (nothing overlaps a field launch)
stats one_loop_at_a_time --steps 20 --warmup 5 --pipeline 0 --no-cpu-baseline --no-secondary || exit 1
stats sequential --steps 20 --warmup 3 --pipeline 0 --group-frames 1 --no-cpu-baseline --no-secondary || exit 1
echo "stats done"
fi
if want pmc; then
# PMC: the static frame, 4 copies per loop (the default frame-group size), one loop at a time: 1 count + 1 warm-up + 2 timed + 2 latency = 6 loops
B="python3 bench.py --static-frame --steps 8 --warmup 1 --pipeline 0 --no-cpu-baseline --no-secondary --min-timed-s 0"
$B > $OUT/bench_pmc_command.json 2>/dev/null || exit 1
timeout -k 10 1000 python3 tools/pmc_passes.py $OUT/pmc_field $OUT/pmc_field_raw.json \
  --set A=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS \
  --set B=SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_INSTS_MFMA,SQ_ACTIVE_INST_VALU,SQ_THREAD_CYCLES_VALU,SQ_INSTS_SALU,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS \
  --set C=SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_VALU_MFMA_COEXEC_CYCLES,SQ_ACTIVE_INST_SCA \
  --set D=GRBM_GUI_ACTIVE,GRBM_COUNT --set E=FETCH_SIZE --set F=WRITE_SIZE \
  --set G=TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum,TCC_READ_sum --set H=TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum,TCP_TOTAL_ACCESSES_sum,TCP_TA_DATA_STALL_CYCLES_sum \
  --kernel k_field --kernel k_composite_march --kernel k_march_rays \
  --note "static frame x 4 per loop, one loop at a time" -- $B || exit 1
python3 tools/field_pmc_summary.py $OUT/pmc_field_raw.json $OUT/bench_pmc_command.json 6 $OUT/field_pmc_summary.json || exit 1
echo "pmc done"
fi
if want grid; then
python3 tools/grid_bwd_speed.py > $OUT/grid_bwd_speed.txt 2>/dev/null || exit 1
timeout -k 10 900 python3 tools/pmc_passes.py $OUT/pmc_grid $OUT/pmc_grid_raw.json \
  --set A=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_VMEM_RD \
  --set D=GRBM_GUI_ACTIVE,GRBM_COUNT --set E=FETCH_SIZE --set F=WRITE_SIZE \
  --set G=TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum,TCC_READ_sum,TCC_ATOMIC_sum,TCC_WRITE_sum --set H=TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum,TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum,TCP_TCC_WRITE_REQ_sum \
  --set I=TA_TA_BUSY_sum,TA_BUSY_avr,TA_ADDR_STALLED_BY_TC_CYCLES_sum,TA_FLAT_READ_WAVEFRONTS_sum,TA_FLAT_ATOMIC_WAVEFRONTS_sum \
  --kernel k_grid_fwd --kernel k_grid_bwd \
  --note "tools/grid_bwd_speed.py: forward + backward of the stand-alone grid op, 9 000 / 262 144 ray-ordered and 2 097 152 uniform points, fp16 and fp32" -- python3 tools/grid_bwd_speed.py || exit 1
echo "grid pmc done"
fi
if want fp32; then
stats fp32 --fp32 --steps 20 --warmup 5 --no-cpu-baseline || exit 1
python3 tools/field_f32_speed.py > $OUT/field_f32_speed.txt 2>/dev/null || exit 1
echo "fp32 done"
fi
if want pmc32; then
# the fp32 fused field kernel under counters: static frame, one loop at a time
B32="python3 bench.py --fp32 --static-frame --steps 4 --warmup 1 --pipeline 0 --no-cpu-baseline --min-timed-s 0"
$B32 > $OUT/bench_pmc32_command.json 2>/dev/null || exit 1
timeout -k 10 900 python3 tools/pmc_passes.py $OUT/pmc_field32 $OUT/pmc_field_f32_raw.json \
  --set A=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS \
  --set B=SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_INSTS_MFMA,SQ_ACTIVE_INST_VALU,SQ_THREAD_CYCLES_VALU,SQ_INSTS_SALU,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS \
  --set D=GRBM_GUI_ACTIVE,GRBM_COUNT --set E=FETCH_SIZE --set F=WRITE_SIZE \
  --kernel k_field_f32 --note "bench.py --fp32, static frame, one loop at a time" -- $B32 || exit 1
echo "pmc32 done"
fi
if want cpu800; then
python3 bench.py --steps 20 --warmup 5 --no-secondary --cpu-baseline-side 800 > $OUT/bench_cpu_baseline_800.json 2>/dev/null || exit 1
echo "cpu 800 done"
fi
if want modes; then
python3 bench.py --mode train --steps 300 --warmup 10 > $OUT/bench_train.json 2>/dev/null || exit 1
SDN_DETERMINISTIC=1 python3 bench.py --mode train --steps 300 --warmup 10 > $OUT/bench_train_deterministic.json 2>/dev/null || exit 1
python3 bench.py --mode seald --steps 20 > $OUT/bench_seald.json 2>/dev/null || exit 1
python3 bench.py --mode seald-train --steps 50 > $OUT/bench_seald_train.json 2>/dev/null || exit 1
python3 bench.py --mode density --steps 8 > $OUT/bench_density.json 2>/dev/null || exit 1
python3 bench.py --steps 384 --scene lego --no-cpu-baseline --no-secondary > $OUT/bench_lego.json 2>/dev/null || exit 1
echo "modes done"
fi
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls $OUT | head -60
