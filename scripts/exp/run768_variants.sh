for v in base w12_22 w12_31; do
  if [ $v = base ]; then unset RABITQ_HIP_SO; else export RABITQ_HIP_SO=$PWD/build_exp/lib_$v.so; fi
  timeout -k 10 300 python bench.py --no-secondary --dim 768 --batch 32768 --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --small-batch 0 --gt-queries 100 > gpurun_out/b_768_$v.json 2> gpurun_out/b_768_$v.err || { echo "FAILED $v"; tail -5 gpurun_out/b_768_$v.err; break; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/b_768_$v.json").read().strip().splitlines()[-1])
print("$v", j["value"], j["ms_per_step"], j["recall_at_10"], j["kernel_ms_per_step"]["scan_matrix"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"])
PY
done
