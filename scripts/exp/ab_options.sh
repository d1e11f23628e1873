# A/B of engine options on ONE box, alternating runs: bash scripts/exp/ab_options.sh "tag1:--option a=1 --option b=2" "tag2:..."
for rep in 1 2; do for spec in "$@"; do
  tag=${spec%%:*}; opts=${spec#*:}
  timeout -k 10 300 python bench.py --no-secondary --steps 5 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 100 $opts > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "FAILED $tag"; tail -5 gpurun_out/ab_$tag.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
print("$tag", j["value"], j["ms_per_step"], j["recall_at_10"], "frac", j["roofline"]["frac"], "exact_rate", j.get("matrix_exact_path_rate"), "rerank/q", round(j.get("rerank_candidates_per_query", 0)), "retries", j.get("retries"), {k: v for k, v in j["kernel_ms_per_step"].items()})
PY
done; done
