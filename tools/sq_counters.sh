#!/bin/bash
# SQ occupancy / issue counters of the dominant kernels (one PMC pass each, kernel-trace only):
#   tools/sq_counters.sh r2
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r2}
mkdir -p gpurun_out/$R
LEAN="--no-cpu-baseline --no-fresh-leg --rollout-steps 0 --steps 48 --warmup 16 --repeats 1 --no-kernel-events"
for C in c3 c2; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d gpurun_out/$R/sq_$C -- python3 bench.py --config $C $LEAN > gpurun_out/$R/sq_$C.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$R/sq2_$C -- python3 bench.py --config $C $LEAN > gpurun_out/$R/sq2_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ.get("R", "r2")
for d in sorted(glob.glob(f"gpurun_out/*/sq*_c?")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[(k, row["Counter_Name"])] += 1
    print("==", d)
    for k, v in acc.items():
        print(" ", k, {c: round(x / max(n[(k, c)], 1)) for c, x in v.items()}, "launches", max(n[(k, c)] for c in v))
PY
