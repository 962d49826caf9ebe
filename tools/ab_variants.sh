#!/bin/bash
# A/B of library variants on ONE box: tools/ab_variants.sh "<shape indices>" variant.so ...   (shape indices of tools/shape_bench.py)
# Each variant is a build of csrc/groove_hip.hip with other -D tile-rule macros, selected through GT_LIB_PATH; two interleaved rounds.
cd ${GRAFT_REPO_ROOT:-.}
shapes=$1; shift
for round in 1 2; do
  for so in default "$@"; do
    for i in $shapes; do
      if [ "$so" = default ]; then r=$(python tools/shape_bench.py --only $i --steps 60 2>/dev/null | tail -1)
      else r=$(GT_LIB_PATH=$PWD/$so python tools/shape_bench.py --only $i --steps 60 2>/dev/null | tail -1); fi
      echo "$so | $r"
    done
  done
done
