# configs[3] (100M x 768, batch 32768) under engine options: bash scripts/exp/ab_768_options.sh "split_rows=0" "split_rows=1" ...
for rep in 1 2; do for o in "$@"; do
  tag=$(echo "$o" | tr '=, ' '___')
  timeout -k 10 500 python bench.py --no-secondary --dim 768 --batch 32768 --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 50 --option $o > gpurun_out/b768_$tag.json 2> gpurun_out/b768_$tag.err || { echo "FAILED $o"; tail -3 gpurun_out/b768_$tag.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/b768_$tag.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_step"]
print("$o", j["value"], j["ms_per_step"], j["recall_at_10"], "| scan_matrix", k["scan_matrix"], "early", round(k["scan"]-k["scan_matrix"],3), "rerank", k["rerank"], "coarse", k["coarse"], "prep", k["prep"], "cand/q", round(j["rerank_candidates_per_query"],1), "rejects/q", j.get("rerank_shadow_rejects_per_query"), "build s", j["build_seconds"])
PY
done; done
