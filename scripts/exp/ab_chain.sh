#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of the matrix-core scan's block mapping: arguments "chain:xcd" (options scan_chain, scan_xcd)
for cx in "$@"; do
  c=${cx%%:*}; x=${cx##*:}
  timeout -k 10 300 python bench.py --no-secondary --steps 8 --warmup 2 --no-cpu-baseline --no-two-in-flight --small-batch 0 --gt-queries 50 --option scan_chain=$c --option scan_xcd=$x $AB_EXTRA > gpurun_out/ab_chain_${c}_$x.json 2> gpurun_out/ab_chain_${c}_$x.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_chain_${c}_$x.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("scan_chain=$c scan_xcd=$x", "q/s", round(d["value"]), "ms/step", d["ms_per_step"], "scan_matrix", k["scan_matrix"], "launch", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "recall", d["recall_at_10"])
PY
done
