// Translation unit of the sequence-resident backward kernels (gt_seq.h): every instantiation + its launcher.
#define GT_SEQ_TU_BWD
#include "gt_seq.h"

// kernel<DP, HDC, EXACT>: d_model class 32 / 64 / 128, head-dim class 0 (< 16) / 16 / 32 / 64, d_model == DP
#define GT_SEQ_LAUNCH_HD(K, DP, EX, hc, grid, block, s, a)                              \
  if ((hc) == 0) gt_launch(K<DP, 0, EX, false>, grid, block, s, a);                      \
  else if ((hc) == 16) gt_launch(K<DP, 16, EX, false>, grid, block, s, a);               \
  else if ((hc) == 32 || (DP) == 32) gt_launch(K<DP, 32, EX, false>, grid, block, s, a); \
  else gt_launch(K<(DP) == 32 ? 64 : DP, 64, EX, false>, grid, block, s, a);
// the SPLIT kernels (two workgroups per sequence, one launch per phase): d_model 128 or 32 exactly
#define GT_SEQ_LAUNCH_SPLIT(K, dm, hc, grid, block, s, a)                         \
  if ((dm) == 32) {                                                                \
    if ((hc) == 0) gt_launch(K<32, 0, true, true>, grid, block, s, a);             \
    else if ((hc) == 16) gt_launch(K<32, 16, true, true>, grid, block, s, a);      \
    else gt_launch(K<32, 32, true, true>, grid, block, s, a);                      \
  } else if ((hc) == 0) gt_launch(K<128, 0, true, true>, grid, block, s, a);       \
  else if ((hc) == 16) gt_launch(K<128, 16, true, true>, grid, block, s, a);       \
  else if ((hc) == 32) gt_launch(K<128, 32, true, true>, grid, block, s, a);       \
  else gt_launch(K<128, 64, true, true>, grid, block, s, a);
#define GT_SEQ_LAUNCH_DP(K, DP, dm, hc, grid, block, s, a)                                                  \
  { if ((dm) == (DP)) { GT_SEQ_LAUNCH_HD(K, DP, true, hc, grid, block, s, a) } else { GT_SEQ_LAUNCH_HD(K, DP, false, hc, grid, block, s, a) } }
#define GT_SEQ_DISPATCH(K, dm, hc, grid, block, s, a)                         \
  if ((dm) <= 32) GT_SEQ_LAUNCH_DP(K, 32, dm, hc, grid, block, s, a)           \
  else if ((dm) <= 64) GT_SEQ_LAUNCH_DP(K, 64, dm, hc, grid, block, s, a)      \
  else GT_SEQ_LAUNCH_DP(K, 128, dm, hc, grid, block, s, a)
void gt_seq_launch_bwd(const SeqArgs& a, int d_model, int hc, bool split, unsigned nblocks, hipStream_t s, bool quad) {
  const dim3 grid(nblocks), block(GT_SEQ_NT);
  // QUAD: phase 0 with four workgroups per sequence (d_model 128; no attention in that phase, so one head-dim class serves all)
  if (quad) { gt_launch(seq_bwd_kernel<128, 32, true, true, true>, grid, block, s, a); return; }
  if (split && d_model == 64) { gt_seq_launch_bwd64(a, hc, nblocks, s); return; }      // (a translation unit of its own: groove_seq64.hip)
  if (split) { GT_SEQ_LAUNCH_SPLIT(seq_bwd_kernel, d_model, hc, grid, block, s, a) }
  else { GT_SEQ_DISPATCH(seq_bwd_kernel, d_model, hc, grid, block, s, a) }
}
void gt_seq_launch_tail(const SeqArgs& a, unsigned nblocks, hipStream_t s) { gt_launch(seq_tail_kernel, dim3(nblocks), dim3(GT_SEQ_NT), s, a); }
void gt_seq_launch_fb(const SeqArgs& a, unsigned nblocks, hipStream_t s) { gt_launch(seq_fb_kernel<32>, dim3(nblocks), dim3(GT_SEQ_NT), s, a); }
