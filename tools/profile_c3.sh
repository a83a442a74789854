set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r1_bench_c3.json 2> gpurun_out/r1_bench_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3_fused -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_c3_fused.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_c3_write -- python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-kernel-events > gpurun_out/pmc_c3_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_c3_fetch -- python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-kernel-events > gpurun_out/pmc_c3_fetch.log 2>&1
tail -n1 gpurun_out/r1_bench_c3.json | cut -c1-1500
find gpurun_out/prof_c3_fused gpurun_out/pmc_c3_write gpurun_out/pmc_c3_fetch -type f | head -20
