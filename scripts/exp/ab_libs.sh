# A/B of library variants on the SAME box: build_exp/lib_<tag>.so, alternating runs
for rep in 1 2; do for v in "$@"; do
  RABITQ_HIP_SO=$PWD/build_exp/lib_$v.so timeout -k 10 300 python bench.py --no-secondary --steps 5 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 100 $AB_BENCH_ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "FAILED $v"; tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_step"]
print("$v", j["value"], j["ms_per_step"], j["recall_at_10"], "frac", j["roofline"]["frac"], "| scan_matrix", k["scan_matrix"], "early", round(k["scan"]-k["scan_matrix"],3), "rerank", k["rerank"], "coarse", k["coarse"], "prep", k["prep"], "group", k["group"], "replay", k["replay"])
PY
done; done
