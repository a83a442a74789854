#!/bin/bash
# Where does the streaming (nt) store policy start to pay?  Same box, default (sc1) vs stamps_tmp/lib_sc1nt.so.
for spec in "c3 4096" "c3 8192" "c3 16384" "c3 65536" "c4 2048" "c4 4096" "c4 8192" "c4 16384" "c5 1024" "c5 2048" "c5 4096" "c5 8192"; do
  set -- $spec; C=$1; B=$2; ST=160; [ $C = c5 ] && ST=48
  for L in default stamps_tmp/lib_sc1nt.so; do
    if [ $L = default ]; then python bench.py --config $C --envs $B --steps $ST --no-cpu-baseline 2>/dev/null | tail -n1 > /tmp/x.json
    else PCBENV_LIB=$GRAFT_REPO_ROOT/$L python bench.py --config $C --envs $B --steps $ST --no-cpu-baseline 2>/dev/null | tail -n1 > /tmp/x.json; fi
    python -c "import json; b=json.load(open('/tmp/x.json')); r=b['roofline']; print('$C envs $B $L', 'MB/launch', round(r['algorithmic_bytes_per_env_step']*r['units_per_launch']/1e6), round(b['value']/1e6,2), r['frac'])"
  done
done
