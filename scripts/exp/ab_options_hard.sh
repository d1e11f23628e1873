# A/B of engine options on the hard distribution, ONE box, alternating runs: bash scripts/exp/ab_options_hard.sh "tag1:--option a=1" "tag2:..."
for rep in 1 2; do for spec in "$@"; do
  tag=${spec%%:*}; opts=${spec#*:}
  timeout -k 10 400 python bench.py --no-secondary --distribution hard --steps 3 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 100 $opts > gpurun_out/abh_$tag.json 2> gpurun_out/abh_$tag.err || { echo "FAILED $tag"; tail -5 gpurun_out/abh_$tag.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/abh_$tag.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_step"]
print("$tag", j["value"], j["ms_per_step"], j["recall_at_10"], "exact_rate", j.get("matrix_exact_path_rate"), "rerank/q", round(j.get("rerank_candidates_per_query", 0)), "retries", j.get("retries"), "| scan_matrix", k["scan_matrix"], "early", round(k["scan"]-k["scan_matrix"],2), "group", k["group"], "rerank", k["rerank"], "sort", k["sort"], "replay", k["replay"])
PY
done; done
