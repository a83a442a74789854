#!/bin/bash
# Same-box sweep of resident workgroups per CU (LDS padding through PCBENV_LDS_MIN, read only by a -DPCBENV_EXPERIMENTS build:
#   python rl-environment-for-component-placement_amd/build.py -DPCBENV_EXPERIMENTS ab/libx.so; PCBENV_LIB=.../ab/libx.so): tools/occupancy_experiment.sh <config>
C=${1:-c3}
for L in 0 13312 16384 20480 26624 32768 40960 54272; do
  PCBENV_LDS_MIN=$L python bench.py --config $C --no-cpu-baseline 2>/dev/null | tail -n1 > /tmp/occ.json
  python -c "import json; b=json.load(open('/tmp/occ.json')); print('$C lds_min $L', round(b['value']/1e6,2), b['ms_per_step'], b['roofline']['frac'])"
done
