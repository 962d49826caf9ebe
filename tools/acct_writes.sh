#!/bin/bash
# Where do the bytes a forward launch of the headline step WRITES go?  (VERDICT r05 #4 i)  Run on the GPU box through gpurun, after
#   csrc/build.sh; for m in 1 2 3 4; do tools/build_variant.sh acct$m groove_seq_fwd,groove_seq_bwd -DGT_SEQ_ACCT=$m; done
# One rocprofv3 --pmc WRITE_SIZE pass (and one FETCH_SIZE pass) of the eager bench per build: the shipped library, and the diagnostic builds
# that leave one class of stores out (gt_seq.h GT_SEQ_ACCT: 1 = stores saved for the backward alone, 2 = the pair exchange's re-zeroing,
# 3 = both, 4 = no pair-exchange stores at all).  Their results are WRONG by construction; only the counters are read.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/acct; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GT_XCHG_STRICT=0
for v in hip acct1 acct2 acct3 acct4; do
  export GT_LIB_PATH=$R/transformergrooveinfilling_amd/lib/libgroove_$v.so
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/$v/$c -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph > $O/$v.$c.log 2>&1
  done
done
cd $R && python3 tools/acct_writes.py $O | tee $O/summary.txt
