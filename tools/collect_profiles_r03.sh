#!/bin/bash
# Collects round 3's measurement artefacts on the GPU box into gpurun_out/profiles_r03/ (copied into profiles/ afterwards).
# rocprofv3: the profiled program comes right after `--`; counters in their own passes (with --kernel-trace only).
set -o pipefail
OUT=gpurun_out/profiles_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -o p -- python3 bench.py "$@" > $OUT/bench_${name}_under_rocprof.json 2> $OUT/${name}.err || return 1
  cp $(find $OUT/tmp_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/tmp_$name
}
PART=${1:-all}
if [ "$PART" = "groups" ]; then rm -f $OUT/frame_groups_one_gpu.txt; fi
if [ "$PART" != "2" ] && [ "$PART" != "groups" ]; then
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1; echo "bench default done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>/dev/null || exit 1; echo "driver cmd done"
python3 bench.py --steps 20 --pipeline 0 --group-frames 1 --no-cpu-baseline > $OUT/bench_sequential.json 2>/dev/null || exit 1
python3 bench.py --steps 384 --scene lego --no-cpu-baseline --no-secondary > $OUT/bench_lego.json 2>/dev/null || exit 1
python3 bench.py --field ops --steps 5 --warmup 1 --static-frame --no-cpu-baseline > $OUT/bench_ops_f16.json 2>/dev/null || exit 1
echo "bench lines done"
fi
if [ "$PART" != "2" ]; then
for cfg in "--steps 384 --group-frames 1" "--steps 384 --group-frames 2" "--steps 384 --group-frames 4" "--steps 384 --group-frames 8" \
           "--emulate-rank-of 8 --group-frames 16 --steps 384" "--emulate-rank-of 8 --group-frames 8 --steps 384" "--emulate-rank-of 8 --group-frames 10 --steps 20 --warmup 5" \
           "--emulate-rank-of 4 --group-frames 16 --steps 384" "--emulate-rank-of 4 --group-frames 10 --steps 20 --warmup 5" "--emulate-rank-of 2 --group-frames 8 --steps 384" \
           "--emulate-rank-of 2 --group-frames 10 --steps 20 --warmup 5" "--steps 20 --warmup 5"; do
  python3 bench.py $cfg --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(sys.argv[1], '| ms/frame', round(d['ms_per_step'],4), 'points/s', '%.4g'%d['value'], 'repeats', d['repeats'], 'exclusive frac', round(r.get('frac',0),4), 'whole-job', round(r.get('whole_job_mfma_frac',0),4), 'rays/loop', d['config']['rays_per_loop_on_this_gpu'])" "$cfg" >> $OUT/frame_groups_one_gpu.txt || exit 1
done
echo "frame groups done"
fi
if [ "$PART" = "groups" ]; then cat $OUT/frame_groups_one_gpu.txt; exit 0; fi
if [ "$PART" != "2" ]; then
stats default --no-cpu-baseline --no-secondary || exit 1
stats sequential --steps 20 --warmup 3 --pipeline 0 --group-frames 1 --no-cpu-baseline --no-secondary || exit 1
stats train --mode train --steps 100 --warmup 10 || exit 1
echo "stats done"
fi
if [ "$PART" = "1" ]; then ls $OUT; exit 0; fi
python3 bench.py --mode train --steps 300 --warmup 10 > $OUT/bench_train.json 2>/dev/null || exit 1
python3 bench.py --mode train --steps 300 --warmup 10 --train-overlap 0 --train-prefetch 0 > $OUT/bench_train_no_overlap_no_prefetch.json 2>/dev/null || exit 1
python3 bench.py --mode seald --steps 20 > $OUT/bench_seald.json 2>/dev/null || exit 1
python3 bench.py --mode seald-train --steps 50 > $OUT/bench_seald_train.json 2>/dev/null || exit 1
python3 bench.py --mode density --steps 8 > $OUT/bench_density.json 2>/dev/null || exit 1
echo "modes done"
# PMC: the static frame, 4 copies per loop (the default frame-group size), one loop at a time: 1 count + 1 warm-up + 2 timed + 2 latency = 6 loops
B="python3 bench.py --static-frame --steps 8 --warmup 1 --pipeline 0 --no-cpu-baseline --no-secondary --min-timed-s 0"
$B > $OUT/bench_pmc_command.json 2>/dev/null || exit 1
timeout -k 10 1000 python3 tools/pmc_passes.py $OUT/pmc_field $OUT/pmc_field_raw.json \
  --set A=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS \
  --set B=SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_INSTS_MFMA,SQ_ACTIVE_INST_VALU,SQ_THREAD_CYCLES_VALU,SQ_INSTS_SALU,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS \
  --set C=SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_VALU_MFMA_COEXEC_CYCLES,SQ_ACTIVE_INST_SCA \
  --set D=GRBM_GUI_ACTIVE,GRBM_COUNT --set E=FETCH_SIZE --set F=WRITE_SIZE \
  --kernel k_field_f16 --kernel k_composite_march --kernel k_march_rays \
  --note "static frame x 4 per loop, one loop at a time" -- $B || exit 1
python3 tools/field_pmc_summary.py $OUT/pmc_field_raw.json $OUT/bench_pmc_command.json 6 $OUT/field_pmc_summary.json || exit 1
echo "pmc done"
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls $OUT | head -60
