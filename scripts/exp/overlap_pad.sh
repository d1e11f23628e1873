# experiment: does leaving room on the CUs (fewer matrix-core scan blocks per CU, via padded dynamic LDS) let a second batch overlap?
for pad in 0 50 90; do
  RQ_EXP_MFMA_LDS_PAD_KB=$pad timeout -k 10 300 python bench.py --no-secondary --steps 6 --warmup 2 --no-cpu-baseline --small-batch 0 --gt-queries 100 > gpurun_out/ov_$pad.json 2> gpurun_out/ov_$pad.err || { echo FAILED $pad; tail -3 gpurun_out/ov_$pad.err; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/ov_$pad.json").read().strip().splitlines()[-1])
print("pad $pad KB: one at a time", j["value"], "scan_matrix", j["kernel_ms_per_step"]["scan_matrix"], "| two in flight", j["two_batches_in_flight"]["value"], j["two_batches_in_flight"]["ms_per_step"])
PY
done
