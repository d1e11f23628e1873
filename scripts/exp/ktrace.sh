#!/bin/bash
# per-LAUNCH times of the last query batch of a short bench run (rocprofv3 --kernel-trace), in launch order
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
K=$PWD/gpurun_out/ktrace; rm -rf $K
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $K -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-secondary --no-batch-sweep --small-batch 0 --gt-queries 100 "$@" \
    > gpurun_out/bench_ktrace.json 2> gpurun_out/bench_ktrace.log
find $K -name "*.db" -delete 2>/dev/null
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/ktrace/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last batch: from the last rotate_mfma_kernel launch on
idx = [i for i, r in enumerate(rows) if "rotate_mfma_kernel" in r["Kernel_Name"]]
# the breakdown batch runs with profiling events; take the batch before it if there are several
start = idx[-2] if len(idx) >= 2 else idx[-1]
end = idx[-1] if len(idx) >= 2 else len(rows)
t0 = int(rows[start]["Start_Timestamp"])
tot = 0
for r in rows[start:end]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  {d:9.1f} us  grid {r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size","")}  {r["Kernel_Name"][:80]}')
print("sum of kernel times", tot, "us; span", (int(rows[end - 1]["End_Timestamp"]) - t0) / 1e3)
PY
