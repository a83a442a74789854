#!/bin/bash
# Same-box A/B of alternative builds of libpcbenv.so: tools/ab_libs.sh <config> lib1.so lib2.so ... (default build first)
C=$1; shift
for L in default "$@"; do
  for rep in 1 2; do
    if [ $L = default ]; then python bench.py --config $C --no-cpu-baseline 2>/dev/null | tail -n1 > /tmp/ab.json
    else PCBENV_LIB=$GRAFT_REPO_ROOT/$L python bench.py --config $C --no-cpu-baseline 2>/dev/null | tail -n1 > /tmp/ab.json; fi
    python -c "import json; b=json.load(open('/tmp/ab.json')); print('$C $L', round(b['value']/1e6,2), b['ms_per_step'], b['roofline']['frac'])"
  done
done
