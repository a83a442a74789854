#!/bin/bash
# Batch-size scaling of the fused loop on one GPU: tools/batch_scaling.sh > gpurun_out/scaling.txt
for C in c3 c2; do
  for B in 64 1024 4096 16384 65536; do
    ST=320; [ $B -ge 16384 ] && ST=96
    python bench.py --config $C --envs $B --steps $ST --no-cpu-baseline 2>/dev/null | tail -n1 > /tmp/sc.json
    python -c "import json; b=json.load(open('/tmp/sc.json')); print('$C envs $B', round(b['value']/1e6,1), 'M', round(b['ms_per_step']*1e3,1), 'us', b['roofline']['frac'])"
  done
done
