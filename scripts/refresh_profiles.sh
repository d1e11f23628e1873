#!/bin/bash
# Run ON THE GPU BOX (gpurun --timeout 1200 -- 'bash scripts/refresh_profiles.sh'): the rocprofv3 passes and the
# bench run whose summaries scripts/make_profiles.py turns into profiles/*.  Counters in their own pass.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
K=$PWD/gpurun_out/kstats; rm -rf $K
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $K -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-secondary \
    > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.log
echo "kernel trace done"
F=$PWD/gpurun_out/pmc_fetch; rm -rf $F
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $F -- python3 bench.py --steps 1 --warmup 1 \
    --no-cpu-baseline --no-two-in-flight --no-secondary --gt-queries 10 --small-batch 0 > /dev/null 2> gpurun_out/pmc_fetch.log
echo "pmc pass done"
timeout -k 10 900 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log
timeout -k 10 600 python3 bench.py --two-in-flight --no-cpu-baseline --no-secondary --gt-queries 100 > gpurun_out/bench_two_in_flight.json 2> gpurun_out/bench_two_in_flight.log
RQ_PROFILE_OUT=$PWD/gpurun_out/profiles python3 scripts/make_profiles.py r03 > gpurun_out/make_profiles.log 2>&1 || tail -5 gpurun_out/make_profiles.log
find gpurun_out/kstats gpurun_out/pmc_fetch -name "*.db" -delete 2>/dev/null || true
tail -c 600 gpurun_out/bench_final.json
