#!/bin/bash
# Round-end evidence on the GPU box: bench JSON, rocprofv3 kernel stats and PMC traffic (separate passes) per config.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r1}
mkdir -p gpurun_out/$R
for C in c3 c4 c5 c2; do
  ST=320; [ $C = c5 ] && ST=64
  python bench.py --config $C --steps $ST > gpurun_out/$R/${C}_bench.json 2> gpurun_out/$R/${C}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_$C -- python3 bench.py --config $C --steps $ST --no-cpu-baseline > gpurun_out/$R/prof_$C.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmcw_$C -- python3 bench.py --config $C --steps 48 --warmup 16 --no-cpu-baseline --no-kernel-events > gpurun_out/$R/pmcw_$C.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmcf_$C -- python3 bench.py --config $C --steps 48 --warmup 16 --no-cpu-baseline --no-kernel-events > gpurun_out/$R/pmcf_$C.log 2>&1
  echo "== $C"; tail -n1 gpurun_out/$R/${C}_bench.json | cut -c1-400
done
python bench.py --config c3 --reward both --no-cpu-baseline > gpurun_out/$R/c3_both_bench.json 2>/dev/null
python bench.py --config c3 --loop explicit --no-cpu-baseline > gpurun_out/$R/c3_explicit_bench.json 2>/dev/null
python bench.py --config c3 --incremental --no-cpu-baseline > gpurun_out/$R/c3_incremental_bench.json 2>/dev/null
python bench.py --config c4 --incremental --no-cpu-baseline > gpurun_out/$R/c4_incremental_bench.json 2>/dev/null
