#!/bin/bash
# per-kernel time of a large-batch step (run on the GPU box): rocprofv3 --kernel-trace --stats of a short bench run
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
K=$PWD/gpurun_out/kstats; rm -rf $K
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $K -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-secondary --no-batch-sweep --small-batch 0 --gt-queries 100 "$@" \
    > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.log
find gpurun_out/kstats -name "*.db" -delete 2>/dev/null
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/kstats/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:32]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f} {float(r["Percentage"]):5.1f}%')
PY
