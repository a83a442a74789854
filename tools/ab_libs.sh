#!/bin/bash
# Same-box A/B of alternative builds of libpcbenv.so: tools/ab_libs.sh <config> [bench args --] lib1.so lib2.so ... (default build first)
C=$1; shift
EXTRA=""
if [[ " $* " == *" -- "* ]]; then while [ "$1" != "--" ]; do EXTRA="$EXTRA $1"; shift; done; shift; fi
for L in default "$@"; do
  for rep in 1 2; do
    if [ $L = default ]; then python bench.py --config $C --no-cpu-baseline --no-fresh-leg --rollout-steps 0 $EXTRA 2>/dev/null | tail -n1 > /tmp/ab.json
    else PCBENV_LIB=$GRAFT_REPO_ROOT/$L python bench.py --config $C --no-cpu-baseline --no-fresh-leg --rollout-steps 0 $EXTRA 2>/dev/null | tail -n1 > /tmp/ab.json; fi
    python -c "import json; b=json.load(open('/tmp/ab.json')); print('$C $EXTRA $L', round(b['value']/1e6,2), b['ms_per_step_repeats']['median'], b['roofline']['frac'])"
  done
done
