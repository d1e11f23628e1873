#!/bin/bash
# per-LAUNCH times of single-query calls (rocprofv3 --kernel-trace of scripts/small_batch_latency.py --batches 1), the last few calls
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
K=$PWD/gpurun_out/ktrace1; rm -rf $K
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $K -- python3 scripts/small_batch_latency.py --batches 1 --reps 12 > gpurun_out/sbl_trace.json 2> gpurun_out/sbl_trace.err
find $K -name "*.db" -delete 2>/dev/null
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/ktrace1/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# single-query calls of the few-launch path: they contain sb_query_kernel; take the 3rd-last such call
idx = [i for i, r in enumerate(rows) if "sb_query_kernel" in r["Kernel_Name"]]
for which in (-8, -7):
    c = idx[which]
    lo = c
    while lo > 0 and int(rows[lo]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 30000: lo -= 1
    hi = c
    while hi + 1 < len(rows) and int(rows[hi + 1]["Start_Timestamp"]) - int(rows[hi]["End_Timestamp"]) < 30000: hi += 1
    t0 = int(rows[lo]["Start_Timestamp"])
    print("call:")
    for r in rows[lo:hi + 1]:
        print(f'  {(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} us  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f} us  {r["Kernel_Name"][:70]}')
    print("  span", (int(rows[hi]["End_Timestamp"]) - t0) / 1e3)
PY
rm -rf $K
