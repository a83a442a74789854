#!/bin/bash
# Instruction-cache counters of k_step with and without the on-device generator running beside it:
#   tools/icache_counters.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ic
A="--config c3 --no-cpu-baseline --no-fresh-leg --rollout-steps 0 --steps 256 --warmup 16 --repeats 1 --no-kernel-events"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ic/replay -- python3 bench.py $A > gpurun_out/ic/replay.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ic/fresh -- python3 bench.py $A --instances device > gpurun_out/ic/fresh.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/ic/replay", "gpurun_out/ic/fresh"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", d)
    for k, v in acc.items():
        if "k_step" in k or "k_gen" in k:
            print(" ", k, {c: (round(sum(x) / len(x)), round(max(x))) for c, x in v.items()}, "launches", max(len(x) for x in v.values()))
PY
