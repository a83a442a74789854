#!/bin/bash
# Round-end evidence on the GPU box: bench JSON, rocprofv3 kernel stats and PMC traffic (separate passes) per config.
#   tools/profile_all.sh r2 [quick | rest]   (quick: the per-config part only; rest: everything after it -- two calls fit gpurun's limit)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r2}
mkdir -p gpurun_out/$R
LEAN="--no-cpu-baseline --no-fresh-leg --no-staggered-leg --rollout-steps 0"   # the headline loop only: one kernel name in the stats
if [ "$2" != rest ]; then
for C in c3 c4 c5 c2; do
  ST=320; [ $C = c5 ] && ST=64
  python bench.py --config $C --steps $ST > gpurun_out/$R/${C}_bench.json 2> gpurun_out/$R/${C}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_$C -- python3 bench.py --config $C --steps $ST $LEAN > gpurun_out/$R/prof_$C.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmcw_$C -- python3 bench.py --config $C --steps 48 --warmup 16 --repeats 1 --no-kernel-events $LEAN > gpurun_out/$R/pmcw_$C.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmcf_$C -- python3 bench.py --config $C --steps 48 --warmup 16 --repeats 1 --no-kernel-events $LEAN > gpurun_out/$R/pmcf_$C.log 2>&1
  echo "== $C"; tail -n1 gpurun_out/$R/${C}_bench.json | cut -c1-300
done
# larger batches: do the bytes reach HBM when a launch writes well beyond the 256 MiB Infinity Cache?
for CB in c3:16384 c3:65536 c4:16384; do
  C=${CB%%:*}; B=${CB##*:}
  python bench.py --config $C --envs $B --steps 96 $LEAN > gpurun_out/$R/${C}_${B}_bench.json 2> gpurun_out/$R/${C}_${B}_bench.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmcw_${C}_$B -- python3 bench.py --config $C --envs $B --steps 32 --warmup 16 --repeats 1 --no-kernel-events $LEAN > gpurun_out/$R/pmcw_${C}_$B.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmcf_${C}_$B -- python3 bench.py --config $C --envs $B --steps 32 --warmup 16 --repeats 1 --no-kernel-events $LEAN > gpurun_out/$R/pmcf_${C}_$B.log 2>&1
  echo "== $C x $B"; tail -n1 gpurun_out/$R/${C}_${B}_bench.json | cut -c1-200
done
fi
[ "$2" = quick ] && exit 0
# the other loops: fresh on-device instances and the persistent rollout (kernel names k_step<..., TRAJ> and k_gen_fill)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_c3_fresh -- python3 bench.py --config c3 --instances device --no-cpu-baseline --rollout-steps 0 --repeats 1 > gpurun_out/$R/prof_c3_fresh.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_c3_rollout -- python3 bench.py --config c3 --steps 32 --repeats 1 --no-cpu-baseline --no-fresh-leg > gpurun_out/$R/prof_c3_rollout.log 2>&1
python bench.py --config c3 --reward both --no-cpu-baseline > gpurun_out/$R/c3_both_bench.json 2>/dev/null
python bench.py --config c3 --reward beam --no-cpu-baseline > gpurun_out/$R/c3_beam_bench.json 2>/dev/null
python bench.py --config c3 --loop explicit --no-cpu-baseline > gpurun_out/$R/c3_explicit_bench.json 2>/dev/null
python bench.py --config c3 --incremental --no-cpu-baseline > gpurun_out/$R/c3_incremental_bench.json 2>/dev/null
python bench.py --config c4 --incremental --no-cpu-baseline > gpurun_out/$R/c4_incremental_bench.json 2>/dev/null
# lock-step vs staggered episode phases: with the terminal list's helper wavefronts (default) and without (terminal_teams = 0)
for C in c3 c4 c5; do python tools/stagger_experiment.py $C 2>&1 | grep stagger | sed "s/^/[helpers on ] /"; python tools/stagger_experiment.py $C 0 2>&1 | grep stagger | sed "s/^/[helpers off] /"; done > gpurun_out/$R/stagger.txt 2>&1
# the trajectory layout with one launch per step (the layout a PPO collect uses): float64 vs compact feature tensors
for C in c3 c4; do python tools/traj_step_experiment.py $C 2>&1 | grep trajectory; done > gpurun_out/$R/trajectory_one_launch_per_step.txt 2>&1
for C in c3 c4; do python tools/launch_times.py $C 2>&1 | grep teams; python tools/launch_times.py $C "" stagger 2>&1 | grep teams; done > gpurun_out/$R/launch_times.txt 2>&1
