#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/profiles_r02/ (copied into profiles/ afterwards).
# rocprofv3: the profiled program comes right after `--`; counters in their own passes (with --kernel-trace only).
set -o pipefail
OUT=gpurun_out/profiles_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -o p -- python3 bench.py "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/${name}.err
  cp $(find $OUT/tmp_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/tmp_$name
}
pmc() {  # name, counters, program...
  local name=$1 counters=$2; shift 2
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT/pmc_$name -o p -- "$@" > $OUT/pmc_${name}.out 2> $OUT/pmc_${name}.err
}
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default done"      # the driver's command: 384 steps, 3 warm-up
python3 bench.py --steps 20 --no-cpu-baseline > $OUT/bench_default_20.json 2>/dev/null
python3 bench.py --steps 20 --pipeline 0 --no-cpu-baseline > $OUT/bench_sequential.json 2>/dev/null
python3 bench.py --steps 20 --static-frame --no-cpu-baseline > $OUT/bench_static_frame.json 2>/dev/null
python3 bench.py --steps 20 --scene lego --no-cpu-baseline > $OUT/bench_lego.json 2>/dev/null
stats default --no-cpu-baseline; echo "stats default done"
stats sequential --steps 20 --warmup 3 --pipeline 0 --no-cpu-baseline
stats train --mode train --steps 100 --warmup 10
stats train_graph --mode train --train-native 0 --steps 30
stats ops_f16 --field ops --steps 5 --warmup 1 --static-frame --no-cpu-baseline
echo "stats done"
# PMC: one static frame rendered 8 times (1 count + 1 warm-up + 2 warm stream + 2 timed + 2 latency)
B="python3 bench.py --steps 2 --warmup 1 --static-frame --no-cpu-baseline"
pmc field_fetch FETCH_SIZE $B
pmc field_write WRITE_SIZE $B
pmc field_sq "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" $B
G="python3 bench.py --field ops --steps 2 --warmup 1 --static-frame --no-cpu-baseline"
pmc grid_fwd_fetch FETCH_SIZE $G
pmc grid_fwd_write WRITE_SIZE $G
pmc grid_bwd_fetch FETCH_SIZE python3 tools/grid_bwd_speed.py
pmc grid_bwd_write WRITE_SIZE python3 tools/grid_bwd_speed.py
echo "pmc done"
python3 tools/grid_bwd_speed.py > $OUT/grid_bwd_speed.txt 2>/dev/null
for cfg in "--emulate-rank-of 8 --group-frames 8 --steps 384" "--emulate-rank-of 8 --group-frames 8 --steps 96" "--emulate-rank-of 8 --group-frames 1 --steps 96" "--emulate-rank-of 8 --group-frames 5 --steps 20" "--emulate-rank-of 4 --group-frames 4 --steps 384" "--emulate-rank-of 2 --group-frames 2 --steps 384" "--steps 384" "--steps 96" "--steps 20"; do
  python3 bench.py $cfg --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '| ms/frame', round(d['ms_per_step'],4), 'latency ms/loop', round(d['latency_ms_one_loop_at_a_time'],3), 'points/s', '%.4g'%d['value'], 'rays/loop', d['config']['rays_per_loop_on_this_gpu'])" "$cfg" >> $OUT/frame_groups_one_gpu.txt
done
python3 bench.py --mode train --steps 200 --warmup 10 > $OUT/bench_train.json 2>/dev/null
python3 bench.py --mode train --train-native 0 --steps 50 > $OUT/bench_train_graph.json 2>/dev/null
python3 tools/ffmlp_speed.py --iters 10 > $OUT/ffmlp_speed.jsonl 2>/dev/null
python3 bench.py --mode seald --steps 20 > $OUT/bench_seald.json 2>/dev/null
python3 bench.py --mode density --steps 8 > $OUT/bench_density.json 2>/dev/null
python3 bench.py --mode seald-train --steps 20 > $OUT/bench_seald_train.json 2>/dev/null
rocprofv3 -L 2>/dev/null | grep -i "SQ_INSTS\|SQ_WAIT\|SQ_ACTIVE\|FETCH_SIZE\|WRITE_SIZE" | head -40 > $OUT/counters_available.txt
# drop the bulky per-dispatch traces, keep the counter tables
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls -la $OUT | head -60
