// Translation unit of the d_model-64 instantiations of the two-workgroups-per-sequence (SPLIT) kernels (gt_seq.h; round 6) and their launchers.
// A code object of its own ON PURPOSE: with these eight kernels inside groove_seq_fwd / groove_seq_bwd the headline step -- whose kernels
// compile to the same instructions either way -- ran 0.4-0.5 % slower (0.1994 vs 0.1986 ms, objects of the two builds mixed and matched on one
// box: the slow-down follows the two translation units' code objects, not the host code).
#define GT_SEQ_TU_64
#include "gt_seq.h"

void gt_seq_launch_fwd64(const SeqArgs& a, int hc, unsigned nblocks, hipStream_t s) {
  const dim3 grid(nblocks), block(GT_SEQ_NT);
  if (hc == 0) gt_launch(seq_fwd_kernel<64, 0, true, true>, grid, block, s, a);
  else if (hc == 16) gt_launch(seq_fwd_kernel<64, 16, true, true>, grid, block, s, a);
  else if (hc == 32) gt_launch(seq_fwd_kernel<64, 32, true, true>, grid, block, s, a);
  else gt_launch(seq_fwd_kernel<64, 64, true, true>, grid, block, s, a);
}
void gt_seq_launch_bwd64(const SeqArgs& a, int hc, unsigned nblocks, hipStream_t s) {
  const dim3 grid(nblocks), block(GT_SEQ_NT);
  if (hc == 0) gt_launch(seq_bwd_kernel<64, 0, true, true>, grid, block, s, a);
  else if (hc == 16) gt_launch(seq_bwd_kernel<64, 16, true, true>, grid, block, s, a);
  else if (hc == 32) gt_launch(seq_bwd_kernel<64, 32, true, true>, grid, block, s, a);
  else gt_launch(seq_bwd_kernel<64, 64, true, true>, grid, block, s, a);
}
