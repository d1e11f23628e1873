# needs the developer build: make -C rabitq_amd/csrc dev
export RABITQ_HIP_SO=$PWD/rabitq_amd/librabitq_hip_dev.so
# matrix-core scan launch time under timing ablations (results are WRONG with most of them; only scan_matrix ms matters)
for o in 64 8256 66 8258 68; do timeout -k 10 300 python bench.py --no-secondary --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --small-batch 0 --gt-queries 10 --option scan_debug=$o > gpurun_out/b_abl$o.json 2> gpurun_out/b_abl$o.err; python - <<PY
import json
j=json.loads(open("gpurun_out/b_abl$o.json").read().strip().splitlines()[-1])
print("scan_debug=$o", "scan_matrix ms", j["kernel_ms_per_step"]["scan_matrix"], "roofline avg launch", j["roofline"]["avg_launch_ms"])
PY
done
