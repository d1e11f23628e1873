#!/bin/bash
# one box: parity of the matrix-core scan tests for each variant library, then the alternating bench A/B (scripts/exp/ab_libs.sh)
# usage: bash scripts/exp/ab_session.sh base v1 v2 ...   (build_exp/lib_<tag>.so)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "$@"; do
  [ $v = base ] && continue
  RABITQ_HIP_SO=$PWD/build_exp/lib_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
     -k "scan_implementations or scan_grid_chunking or ranked_group or degenerate_factors or query_matches_golden or every_option or additive_gate or fuzz_slice" \
     > gpurun_out/ab_parity_$v.log 2>&1 || { echo "PARITY FAILED $v"; tail -20 gpurun_out/ab_parity_$v.log; exit 1; }
  echo "parity $v: $(tail -1 gpurun_out/ab_parity_$v.log)"
done
bash scripts/exp/ab_libs.sh "$@"
