#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a short bench run (arguments = extra bench flags), the per-kernel stats and the
# kernel timeline of the last timed full-batch step -> gpurun_out/tl_<tag>.txt   (usage: timeline2.sh TAG [bench flags])
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
K=$PWD/gpurun_out/kt_$TAG; rm -rf $K
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $K -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-secondary --small-batch 0 --gt-queries 100 "$@" \
    > gpurun_out/tl_$TAG.json 2> gpurun_out/tl_$TAG.log || { tail -5 gpurun_out/tl_$TAG.log; exit 1; }
python3 - "$K" > gpurun_out/tl_$TAG.txt <<'PY'
import csv, glob, sys
K = sys.argv[1]
f = sorted(glob.glob(K + "/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:30]:
    print(f'{r["Name"][:80]:80s} calls {int(r["Calls"]):6d} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f}')
f = sorted(glob.glob(K + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step whose pass starts with the large-batch rotation: walk back from the last finalize_heap_kernel
ends = [i for i, r in enumerate(rows) if "finalize_heap_kernel" in r["Kernel_Name"]]
# pick the last step with > 40 kernels between rotate and finalize (a full batch)
pick = None
for e in reversed(ends):
    j = e
    while j > 0 and "rotate_mfma" not in rows[j]["Kernel_Name"]:
        j -= 1
    if e - j > 30 and (int(rows[e]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) > 5e6:
        pick = (j, e); break
print()
if pick:
    j, e = pick
    t0 = int(rows[j]["Start_Timestamp"]); prev = t0
    agg = {}
    for r in rows[j:e + 2]:
        s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].replace("void ", "")[:56]
        if en - s > 20000:
            print(f"{(s - t0) / 1e3:9.1f}us dur {(en - s) / 1e3:8.1f}us gap {(s - prev) / 1e3:6.1f}  {nm}")
        k = nm.split("(")[0]
        agg[k] = agg.get(k, 0) + (en - s)
        prev = en
    print(f"step wall {(prev - t0) / 1e6:.3f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {k[:60]:60s} {v / 1e6:8.3f} ms")
PY
rm -rf $K
tail -c 300 gpurun_out/tl_$TAG.json | head -c 10 > /dev/null
echo "timeline $TAG done"
