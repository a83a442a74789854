#!/bin/bash
# Same-box A/B of the persistent-rollout leg (trajectory layout): tools/ab_rollout_leg.sh <config> lib1.so ... (default build first)
C=$1; shift
for L in default "$@"; do
  for rep in 1 2 3; do
    if [ $L = default ]; then python bench.py --config $C --no-cpu-baseline --no-fresh-leg --steps 64 --repeats 1 2>/dev/null | tail -n1 > /tmp/abr.json
    else PCBENV_LIB=$GRAFT_REPO_ROOT/$L python bench.py --config $C --no-cpu-baseline --no-fresh-leg --steps 64 --repeats 1 2>/dev/null | tail -n1 > /tmp/abr.json; fi
    python -c "import json; b=json.load(open('/tmp/abr.json')); r=b['rollout']; print('$C $L value', round(b['value']/1e6,1), 'rollout', round(r['value']/1e6,1), r['kernel_ms_per_launch'], r['frac_of_8TBps'])"
  done
done
