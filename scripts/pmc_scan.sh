#!/bin/bash
# SQ counters of the matrix-core scan launch (run on the GPU box through gpurun):
#   gpurun --timeout 1200 -- 'bash scripts/pmc_scan.sh'
# Two separate --pmc passes (counters only with --kernel-trace), summaries under gpurun_out/.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-two-in-flight --no-secondary --no-batch-sweep --gt-queries 10 --small-batch 0 $PMC_EXTRA"  # PMC_EXTRA: e.g. "--dim 768 --batch 32768"
OUT=$PWD/gpurun_out/pmc_mfma; rm -rf $OUT
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $OUT -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmc_mfma.log
OUT2=$PWD/gpurun_out/pmc_mfma2; rm -rf $OUT2
timeout -k 10 500 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD \
    --kernel-trace --output-format csv -d $OUT2 -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmc_mfma2.log
python3 scripts/pmc_scan_summary.py
