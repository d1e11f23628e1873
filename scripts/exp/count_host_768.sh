RABITQ_HIP_SO=$PWD/build_exp/lib_counthost.so timeout -k 10 500 python bench.py --no-secondary --dim 768 --batch 32768 --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 50 > gpurun_out/b768_ch.json 2> gpurun_out/b768_ch.err
python - <<PY
import json
j=json.loads(open("gpurun_out/b768_ch.json").read().strip().splitlines()[-1])
print("host-tier survivors per query", j.get("rerank_shadow_rejects_per_query"), "cand/q", j["rerank_candidates_per_query"], "rerank ms", j["kernel_ms_per_step"]["rerank"])
PY
