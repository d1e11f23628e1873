# A/B of (library, options) pairs on one box: bash scripts/exp/ab_libs_opt.sh "tag:lib:--option a=1" ...
for rep in 1 2; do for spec in "$@"; do
  tag=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; opts=${rest#*:}
  RABITQ_HIP_SO=$PWD/build_exp/lib_$lib.so timeout -k 10 300 python bench.py --no-secondary --steps 5 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 100 $opts > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "FAILED $tag"; tail -3 gpurun_out/ab_$tag.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_step"]
print("$tag", j["value"], j["ms_per_step"], j["recall_at_10"], "frac", j["roofline"]["frac"], "| scan_matrix", k["scan_matrix"], "early", round(k["scan"]-k["scan_matrix"],3), "rerank", k["rerank"], "group", k["group"])
PY
done; done
