#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_rehearse; mkdir -p $OUT
export SDN_REHEARSE_ON_ONE_GPU=1
for n in 2 4; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 20 --warmup 5 > $OUT/b$n.json 2> $OUT/b$n.err || { tail -30 $OUT/b$n.err; exit 1; }
  tail -1 $OUT/b$n.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('n_gpus','steps','repeats','frames_rendered','ms_per_step','value','scaling')})
print('config', d['config'].get('parallelism'), d['config'].get('frames_per_loop'))
print('ranks', json.dumps(d.get('ranks'))[:1500])
print('roofline frac', (d.get('roofline') or {}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"
done
