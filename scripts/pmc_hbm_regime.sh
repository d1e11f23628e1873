#!/bin/bash
# HBM-regime evidence for the scan (run on the GPU box: gpurun --timeout 1100 -- 'bash scripts/pmc_hbm_regime.sh'):
# per workload one plain run (HIP-event timings), one --kernel-trace --stats pass and one --pmc FETCH_SIZE pass
# (counters in their own pass, with --kernel-trace only).  Summaries: scripts/pmc_hbm_summary.py -> profiles/.
set -e
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() {  # tag, args...
    tag=$1; shift
    timeout -k 10 300 python3 scripts/hbm_regime.py "$@" > gpurun_out/hbm_${tag}_plain.json 2> gpurun_out/hbm_${tag}_plain.log
    echo "$tag plain done"
    rm -rf gpurun_out/hbm_${tag}_kt
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/hbm_${tag}_kt -- python3 scripts/hbm_regime.py "$@" \
        > gpurun_out/hbm_${tag}_kt.json 2> gpurun_out/hbm_${tag}_kt.log
    echo "$tag kernel-trace done"
    rm -rf gpurun_out/hbm_${tag}_pmc
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $PWD/gpurun_out/hbm_${tag}_pmc -- python3 scripts/hbm_regime.py "$@" \
        > gpurun_out/hbm_${tag}_pmc.json 2> gpurun_out/hbm_${tag}_pmc.log
    echo "$tag pmc done"
}
run d128 --vectors 100000000 --dim 128 --batches 64,16,1 --reps 20
run d768 --vectors 30000000 --dim 768 --batches 64,16,1 --reps 10
# the big per-dispatch traces are not needed back home: keep the csv files only
find gpurun_out -name '*.db' -delete 2>/dev/null || true
RQ_PROFILE_OUT=$PWD/gpurun_out/profiles python3 scripts/pmc_hbm_summary.py $R > gpurun_out/hbm_summary.log 2>&1 || true
tail -c 3000 gpurun_out/hbm_summary.log
rm -rf gpurun_out/hbm_d128_kt gpurun_out/hbm_d128_pmc gpurun_out/hbm_d768_kt gpurun_out/hbm_d768_pmc   # (only the summaries travel home)
