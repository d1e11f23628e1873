# A/B on ONE box, alternating runs: gate of the matrix-core scan (1 = bf16 rank-5 threshold MFMA, 0 = auto -> additive bound at dim 128)
for rep in 1 2; do for g in 1 0; do
  timeout -k 10 300 python bench.py --no-secondary --steps 5 --warmup 2 --no-cpu-baseline --no-two-in-flight --small-batch 0 --gt-queries 100 --option scan_gate=$g > gpurun_out/ab_gate$g.json 2> gpurun_out/ab_gate$g.err || { echo "FAILED gate $g"; tail -5 gpurun_out/ab_gate$g.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_gate$g.json").read().strip().splitlines()[-1])
print("scan_gate=$g", j["value"], j["ms_per_step"], j["recall_at_10"], "scan_matrix", j["kernel_ms_per_step"]["scan_matrix"], "frac", j["roofline"]["frac"], {k: v for k, v in j["kernel_ms_per_step"].items()})
PY
done; done
