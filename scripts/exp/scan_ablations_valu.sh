# needs the developer build: make -C rabitq_amd/csrc dev
export RABITQ_HIP_SO=$PWD/rabitq_amd/librabitq_hip_dev.so
# the VALU scan stages (scan - scan_matrix) with survivor recording switched off (bit 10: results WRONG)
for o in 0 1024; do timeout -k 10 300 python bench.py --no-secondary --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --small-batch 0 --gt-queries 10 --option scan_debug=$o > gpurun_out/b_ablv_$o.json 2> gpurun_out/b_ablv_$o.err; python - <<PY
import json
j=json.loads(open("gpurun_out/b_ablv_$o.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_step"]
print("scan_debug=$o", "VALU stages ms", round(k["scan"]-k["scan_matrix"],3), "scan_matrix", k["scan_matrix"], "rerank", k["rerank"])
PY
done
