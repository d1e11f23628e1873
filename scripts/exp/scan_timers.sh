# needs the developer build: cycle counters of the matrix-core scan's waves (stderr lines of scan_debug = 256 [+ ablation bits]); extra bench args in TM_ARGS
export RABITQ_HIP_SO=$PWD/rabitq_amd/librabitq_hip_dev.so
for o in ${TM_BITS:-256}; do
timeout -k 10 400 python bench.py --no-secondary --steps 2 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 10 $TM_ARGS --option scan_debug=$o > gpurun_out/b_tm.json 2> gpurun_out/b_tm.err
echo "scan_debug=$o"; grep "scan_mfma" gpurun_out/b_tm.err | tail -2 | cut -c1-400
python - <<PY
import json
j=json.loads(open("gpurun_out/b_tm.json").read().strip().splitlines()[-1])
print("  scan_matrix ms", j["kernel_ms_per_step"]["scan_matrix"], "avg launch", j["roofline"]["avg_launch_ms"])
PY
done
