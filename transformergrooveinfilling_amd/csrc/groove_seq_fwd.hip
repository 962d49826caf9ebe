// Translation unit of the sequence-resident forward kernels (gt_seq.h): every instantiation + its launcher.
#define GT_SEQ_TU_FWD
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
void gt_seq_launch_pack(const SeqArgs& a, unsigned nblocks, hipStream_t s) { gt_launch(seq_pack_kernel, dim3(nblocks), dim3(256), s, a); }
void gt_seq_launch_fwd(const SeqArgs& a, int d_model, int hc, bool split, unsigned nblocks, hipStream_t s, bool quad) {
  const dim3 grid(nblocks), block(GT_SEQ_NT);
  if (quad) {                                  // four workgroups per sequence (d_model 128 exactly): nblocks = 4 x batch
    if (hc == 0) gt_launch(seq_fwd_kernel<128, 0, true, true, true>, grid, block, s, a);
    else if (hc == 16) gt_launch(seq_fwd_kernel<128, 16, true, true, true>, grid, block, s, a);
    else if (hc == 32) gt_launch(seq_fwd_kernel<128, 32, true, true, true>, grid, block, s, a);
    else gt_launch(seq_fwd_kernel<128, 64, true, true, true>, grid, block, s, a);
    return;
  }
  if (split && d_model == 64) { gt_seq_launch_fwd64(a, hc, nblocks, s); return; }      // (a translation unit of its own: groove_seq64.hip)
  if (split) { GT_SEQ_LAUNCH_SPLIT(seq_fwd_kernel, d_model, hc, grid, block, s, a) }
  else { GT_SEQ_DISPATCH(seq_fwd_kernel, d_model, hc, grid, block, s, a) }
}
void gt_seq_launch_update_pack(const SeqArgs& a, int algo, float* params, float* grads, float* m, float* v, int64_t n, const gt_step_state* st,
                               int step_advanced, hipStream_t s) {
  const int64_t frags = (int64_t)a.L * a.kstride / 256;
  SeqUpd u{params, grads, m, v, n, st, algo, step_advanced, (int)((frags + 3) / 4),
           a.xchg >= 0 ? reinterpret_cast<const unsigned*>(a.ws + a.xchg) : nullptr};
  gt_launch(seq_update_pack_kernel, dim3((unsigned)(u.nblk_a + (n / 4 + 255) / 256 + 1)), dim3(256), s, a, u);
}
