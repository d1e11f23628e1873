# A/B of engine options on ONE box: bash scripts/exp/ab_options.sh "" "stage_growth=4" ...
for rep in 1 2; do for o in "$@"; do
  timeout -k 10 300 python bench.py --no-secondary --steps 5 --warmup 2 --no-cpu-baseline --no-two-in-flight --small-batch 0 --gt-queries 100 ${o:+--option $o} > gpurun_out/abo.json 2> gpurun_out/abo.err || { echo "FAILED $o"; tail -3 gpurun_out/abo.err; continue; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/abo.json").read().strip().splitlines()[-1])
print("[$o]", j["value"], j["ms_per_step"], j["recall_at_10"], "rerank/q", round(j["rerank_candidates_per_query"]), {k: v for k, v in j["kernel_ms_per_step"].items() if k in ("scan", "scan_matrix", "rerank", "group", "replay", "sort")})
PY
done; done
