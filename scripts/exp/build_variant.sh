#!/bin/bash
# build_exp/lib_<tag>.so: the library with extra compile flags (kernel experiments A/B-ed on one box by scripts/exp/ab_libs.sh)
# usage: bash scripts/exp/build_variant.sh <tag> "<extra flags>"
set -e
tag=$1; extra=$2
cd "$(dirname "$0")/../../rabitq_amd/csrc"
B=../../build_exp/obj_$tag; mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function $extra"
/opt/rocm/bin/hipcc $FLAGS -c -o $B/rabitq_hip.o rabitq_hip.hip &
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-atomic-optimizer-strategy=None -c -o $B/inst_scan_valu.o inst_scan_valu.hip &
/opt/rocm/bin/hipcc $FLAGS -c -o $B/inst_scan_mfma.o inst_scan_mfma.hip -Rpass-analysis=kernel-resource-usage 2> $B/mfma_usage.txt &
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o ../../build_exp/lib_$tag.so $B/rabitq_hip.o $B/inst_scan_valu.o $B/inst_scan_mfma.o
grep -A8 "scan_mfma_kernelILi2ELi[0-9]*ELb0ELb1E" $B/mfma_usage.txt | grep -E "VGPRs:|Spill|Occupancy|LDS" | tr '\n' ' '; echo
