# needs the developer build: cycle counters of the matrix-core scan's waves at 100M x 768 (stderr lines of scan_debug = 256 [+ ablation bits])
export RABITQ_HIP_SO=$PWD/rabitq_amd/librabitq_hip_dev.so
for o in ${TM_BITS:-256}; do
timeout -k 10 400 python bench.py --no-secondary --dim 768 --batch 32768 --steps 2 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-batch-sweep --small-batch 0 --gt-queries 10 --option scan_debug=$o > gpurun_out/b_tm768.json 2> gpurun_out/b_tm768.err
echo "scan_debug=$o"; grep "scan_mfma timing" gpurun_out/b_tm768.err | tail -1 | cut -c1-400
done
