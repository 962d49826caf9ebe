// Host-visible side of the sequence-resident kernels (gt_seq.h): the argument block and the launch entry points.  The kernels
// are compiled in their own translation units (groove_seq_fwd.hip / groove_seq_bwd.hip) so that the library builds in parallel.
#pragma once
#include "gt_common.h"

#define GT_SEQ_FMAX 512
#ifndef GT_XCHG_SPIN_MAX
#define GT_XCHG_SPIN_MAX (1 << 22)      /* polls of the QUAD pair exchange before a workgroup gives its partner up (seconds); gt_set_xchg_spin_max overrides */
#endif
struct SeqLayerP { int64_t in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b; };
struct SeqLayerW { int64_t qkv, P, ctx, xhat1, rstd1, x1, hact, xhat2, rstd2, xout; };
struct SeqTmp { int64_t dzA, dzAm, dzB, dzBm, dhid, dqkv; };
struct SeqArgs {
  const float* prm; float* ws; const float* pe; const float* xin; float* hvo;
  int B, S, d, F, H, L, hd;
  const gt_step_state* st; uint32_t thr; float dscale;     // dropout (st == nullptr or thr == 0: off)
  SeqLayerP p0; int64_t pstride;                             // layer l: p0.* + l * pstride (encoder layers are laid out uniformly)
  SeqLayerW w0; int64_t wstride;
  SeqTmp t0; int64_t tstride;
  int64_t in_w, in_b, encn_w, encn_b, out_w, out_b;          // parameter offsets of the input layer, final norm, output layer
  int64_t x0, a0, memory, enc_xhat, enc_rstd, dlogits, da0;  // workspace offsets
  int64_t ln_part, ln_part_stride;                           // LayerNorm dgamma/dbeta partials: job j at ln_part + j * stride, [B][2][d]
  int64_t stamps;                                            // diagnostic builds (-DGT_SEQ_STAMPS) only: workspace offset of the stamp buffer
  int64_t pack_f, pack_b, kstride;                           // fragment-ordered weight copies (seq_pack_kernel): workspace offsets, floats per layer
  int64_t dctx;                                              // SPLIT kernels: two [M][d] hand-over buffers of the backward phases (phase p writes
                                                             // buffer p & 1 and reads the other: a fast workgroup must not overwrite rows its
                                                             // partner has yet to read)
  int64_t amask, amask_stride;                               // head_dim-2 attention: keep bits of P, [layer][sequence][head][query] words (-1: absent)
  int64_t xchg;                                              // QUAD forward: the pair-exchange region (8 header granules, then one 16 KB slot per
                                                             // workgroup; zero between launches -- gt_workspace_init); -1: none
  int64_t xchg_b; int fuse_b0;                               // QUAD, fused step: the last forward launch goes on into backward phase 0 (seq_fb_kernel), whose pair exchange uses the region at xchg_b
  int spin_max;                                              // bound of the pair exchange's polling loop (gt_set_xchg_spin_max; default GT_XCHG_SPIN_MAX)
  int quad_pro;                                              // QUAD forward: input layer + in-proj(0) ran as a prologue launch (phase -1)
  int phase;                                                 // SPLIT kernels: which phase this launch runs
  // fused loss (gt_train_step): the launch that runs the output layer also computes the loss terms, d loss / d logits and the
  // step's statistics (loss_y == nullptr: off).  Same arithmetic as loss_kernel<true, true> (gt_loss_elem), one partial per workgroup
  const float* loss_y; float loss_penalty; float* loss_stats; float* loss_part; unsigned* loss_ticket;
  // weight gradients as rider workgroups of the SPLIT backward phases + the tail kernel (gt_seq_wg.h); grd == nullptr: off
  float* grd;                                                // gradient buffer (parameter layout)
  int nseq;                                                  // sequence workgroups of the launch; blocks beyond them are riders
  int wg_accumulate;                                         // 1: add into grd (gt_backward's accumulate), 0: overwrite (grd is zero or dead)
  int ride_last_k;                                           // riders of the last phase cover the tokens [0, ride_last_k); the tail adds the rest
  int out_early;                                             // the output layer's weight gradient was computed by a launch of its own (bucketed backward)
  int tail_phase;                                            // tail kernel: L + 1 (debug launches: another phase's list alone)
  int tail_ksplit;                                           // tail kernel: token chunks per tile (> 1: partial tiles meet in atomics)
  int ln_nwg;                                                // tail kernel: partial rows per LayerNorm instance
  gt_step_state* bump;                                       // tail kernel: advance step / opt_step (fused train step), or nullptr
};
// tiles of one weight-gradient problem in the rider decomposition (32 x 64 outputs per workgroup)
static inline int gt_seq_wg_tiles(int rows, int cols) { return ((rows + 31) / 32) * ((cols + 63) / 64); }
// launchers (one instantiation per d_model class / head-dim class / EXACT / SPLIT); hc = head-dim class 0 (< 16) / 16 / 32 / 64
void gt_seq_launch_pack(const SeqArgs& a, unsigned nblocks, hipStream_t s);
// optimizer update (algo 0 sgd / 1 adam, the arithmetic of gt_misc.h's kernels) + the next step's fragment-ordered weights
void gt_seq_launch_update_pack(const SeqArgs& a, int algo, float* params, float* grads, float* m, float* v, int64_t n, const gt_step_state* st,
                               int step_advanced, hipStream_t s);
void gt_seq_launch_fwd(const SeqArgs& a, int d_model, int hc, bool split, unsigned nblocks, hipStream_t s, bool quad = false);
// floats of the QUAD forward's pair-exchange region for a batch: 8 header granules + 4 x batch slots of 4 x 512 8-byte granules
static inline int64_t gt_seq_xchg_floats(int batch) { return 16 + (int64_t)4 * batch * 4 * 512 * 2; }
// the last forward phase + backward phase 0 of the QUAD schedule in ONE launch (a.fuse_b0 = 1; head-dim class 32 only)
void gt_seq_launch_fb(const SeqArgs& a, unsigned nblocks, hipStream_t s);
void gt_seq_launch_bwd(const SeqArgs& a, int d_model, int hc, bool split, unsigned nblocks, hipStream_t s, bool quad = false);
void gt_seq_launch_tail(const SeqArgs& a, unsigned nblocks, hipStream_t s);
// the SPLIT kernels of d_model 64 (groove_seq64.hip: a code object of their own); hc = head-dim class 0 / 16 / 32 / 64
void gt_seq_launch_fwd64(const SeqArgs& a, int hc, unsigned nblocks, hipStream_t s);
void gt_seq_launch_bwd64(const SeqArgs& a, int hc, unsigned nblocks, hipStream_t s);
