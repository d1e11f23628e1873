#!/bin/bash
# rocprofv3 --kernel-trace --stats of the two secondary workloads (run on the GPU box through gpurun); engine kernels only
# are kept, as profiles/r03_kernel_stats_{hard,768}.csv
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for cfg in "hard --distribution hard" "768 --dim 768 --batch 32768"; do
  set -- $cfg; tag=$1; shift
  rm -rf gpurun_out/ksec_$tag
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/ksec_$tag -- python3 bench.py "$@" --no-secondary --steps 3 --warmup 3 \
      --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 50 > gpurun_out/ksec_$tag.json 2> gpurun_out/ksec_$tag.log
  find gpurun_out/ksec_$tag -name "*.db" -delete
  python3 - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/ksec_$tag/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if not any(t in r["Name"] for t in ("at::native", "Cijk_", "__amd_rocclr", "at::cuda", "rocprim", "hipcub"))]
cfg = json.loads(open("gpurun_out/ksec_$tag.json").read().strip().splitlines()[-1])["config"]["workload"]
with open("gpurun_out/profiles/${R:-r04}_kernel_stats_$tag.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py $* --no-secondary --steps 3 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 50\n")
    o.write(f"# ({cfg}; 3 warm-up + 3 timed + 2 breakdown batches and the streamed build); engine kernels only\n")
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
print("$tag:", len(rows), "kernels")
PY
done
rm -rf gpurun_out/ksec_hard gpurun_out/ksec_768
