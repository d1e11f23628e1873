for B in 49152 65536; do
timeout -k 10 500 python bench.py --no-secondary --dim 768 --batch $B --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 50 > gpurun_out/b768_B$B.json 2> gpurun_out/b768_B$B.err || { echo FAILED $B; tail -5 gpurun_out/b768_B$B.err; }
python - <<PY
import json
j=json.loads(open("gpurun_out/b768_B$B.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_step"]
print($B, j["value"], j["ms_per_step"], j["recall_at_10"], "| scan_matrix", k["scan_matrix"], "early", round(k["scan"]-k["scan_matrix"],3), "rerank", k["rerank"], "coarse", k["coarse"], "prep", k["prep"], "retries", j.get("retries"), "build", j["build"])
PY
done
