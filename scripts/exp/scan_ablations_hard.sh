# needs the developer build: make -C rabitq_amd/csrc dev
export RABITQ_HIP_SO=$PWD/rabitq_amd/librabitq_hip_dev.so
# matrix-core scan on the hard distribution under timing ablations of the exact path (results WRONG with bits 0 / 6;
# 256 = cycle counters of the exact path on stderr)
for o in ${ABL_BITS:-0 1 64 256}; do timeout -k 10 400 python bench.py --no-secondary --distribution hard --steps 2 --warmup 3 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 10 --option scan_debug=$o > gpurun_out/b_ablh_$o.json 2> gpurun_out/b_ablh_$o.err; python - <<PY
import json
j=json.loads(open("gpurun_out/b_ablh_$o.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_step"]
print("scan_debug=$o", "ms/step", j["ms_per_step"], "scan_matrix ms", k["scan_matrix"], "early", round(k["scan"]-k["scan_matrix"],2), "group", k["group"], "rerank", k["rerank"], "sort", k["sort"], "replay", k["replay"], "exact rate", j.get("matrix_exact_path_rate"))
PY
grep "scan_mfma" gpurun_out/b_ablh_$o.err | tail -4 | cut -c1-400
done
