# needs the developer build: make -C rabitq_amd/csrc dev
export RABITQ_HIP_SO=$PWD/rabitq_amd/librabitq_hip_dev.so
# matrix-core scan on the hard distribution under timing ablations of the exact path (results WRONG with bits 0 / 6)
for o in 0 1 64; do timeout -k 10 400 python bench.py --no-secondary --distribution hard --steps 2 --warmup 3 --no-cpu-baseline --no-two-in-flight --small-batch 0 --gt-queries 10 --option scan_debug=$o > gpurun_out/b_ablh_$o.json 2> gpurun_out/b_ablh_$o.err; python - <<PY
import json
j=json.loads(open("gpurun_out/b_ablh_$o.json").read().strip().splitlines()[-1])
print("scan_debug=$o", "ms/step", j["ms_per_step"], "scan_matrix ms", j["kernel_ms_per_step"]["scan_matrix"], "group", j["kernel_ms_per_step"]["group"], "rerank", j["kernel_ms_per_step"]["rerank"])
PY
done
