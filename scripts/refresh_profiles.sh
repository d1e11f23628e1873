#!/bin/bash
# Run ON THE GPU BOX (gpurun --timeout 1200 -- 'bash scripts/refresh_profiles.sh r04'): the rocprofv3 passes and the
# bench runs whose summaries scripts/make_profiles.py turns into profiles/*.  Counters in their own passes (--pmc only with
# --kernel-trace).  Everything lands under gpurun_out/ (scratch); gpurun_out/profiles/ is what gets copied into profiles/.
set -e
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/profiles
K=$PWD/gpurun_out/kstats; rm -rf $K
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $K -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-secondary --no-batch-sweep --small-batch 0 \
    > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.log
echo "kernel trace done"
F=$PWD/gpurun_out/pmc_fetch; rm -rf $F
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $F -- python3 bench.py --steps 1 --warmup 1 \
    --no-cpu-baseline --no-two-in-flight --no-secondary --no-batch-sweep --gt-queries 10 --small-batch 0 > /dev/null 2> gpurun_out/pmc_fetch.log
echo "pmc pass done"
# the secondary workloads: a PMC pass of their own (their roofline.traffic must not borrow the headline workload's bytes)
for cfg in "hard --distribution hard" "768 --dim 768 --batch 32768"; do
  set -- $cfg; tag=$1; shift
  rm -rf gpurun_out/pmc_fetch_$tag
  timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $PWD/gpurun_out/pmc_fetch_$tag -- python3 bench.py "$@" --steps 1 --warmup 3 \
      --no-cpu-baseline --no-two-in-flight --no-secondary --no-batch-sweep --gt-queries 10 --small-batch 0 > gpurun_out/pmc_fetch_$tag.json 2> gpurun_out/pmc_fetch_$tag.log
  RQ_PROFILE_OUT=$PWD/gpurun_out/profiles python3 scripts/pmc_traffic_secondary.py $tag $R > gpurun_out/pmc_traffic_$tag.log 2>&1 || tail -3 gpurun_out/pmc_traffic_$tag.log
  cp gpurun_out/profiles/scan_traffic_$tag.json profiles/ 2>/dev/null || true   # (so that the bench run below quotes it)
  echo "pmc pass $tag done"
done
timeout -k 10 900 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log
RQ_PROFILE_OUT=$PWD/gpurun_out/profiles python3 scripts/make_profiles.py $R > gpurun_out/make_profiles.log 2>&1 || tail -5 gpurun_out/make_profiles.log
find gpurun_out/kstats gpurun_out/pmc_fetch gpurun_out/pmc_fetch_hard gpurun_out/pmc_fetch_768 -name "*.db" -delete 2>/dev/null || true
# only the summaries travel home (gpurun copies back at most 64 MiB): the raw per-dispatch tables stay on the box
rm -rf gpurun_out/kstats gpurun_out/pmc_fetch gpurun_out/pmc_fetch_hard gpurun_out/pmc_fetch_768
tail -c 600 gpurun_out/bench_final.json
