#!/bin/bash
# SQ counters of the matrix-core scan launch of an arbitrary python command (two --pmc passes, counters only with
# --kernel-trace):  gpurun -- 'CMD="scripts/ablate_scan2.py" N=30000000 D=768 B=10000 MODES=0 bash scripts/pmc_cmd.sh'
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/pmc_mfma; rm -rf $OUT
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $OUT -- python3 $CMD > /dev/null 2> gpurun_out/pmc_mfma.log
OUT2=$PWD/gpurun_out/pmc_mfma2; rm -rf $OUT2
timeout -k 10 500 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD \
    --kernel-trace --output-format csv -d $OUT2 -- python3 $CMD > /dev/null 2> gpurun_out/pmc_mfma2.log
python3 scripts/pmc_scan_summary.py
rm -rf $OUT $OUT2
