// Host-visible side of the sequence-resident kernels (gt_seq.h): the argument block and the launch entry points.  The kernels
// are compiled in their own translation units (groove_seq_fwd.hip / groove_seq_bwd.hip) so that the library builds in parallel.
#pragma once
#include "gt_common.h"

#define GT_SEQ_FMAX 512
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
  int phase;                                                 // SPLIT kernels: which phase this launch runs
  // fused loss (gt_train_step): the launch that runs the output layer also computes the loss terms, d loss / d logits and the
  // step's statistics (loss_y == nullptr: off).  Same arithmetic as loss_kernel<true, true> (gt_loss_elem), one partial per workgroup
  const float* loss_y; float loss_penalty; float* loss_stats; float* loss_part; unsigned* loss_ticket;
};
// launchers (one instantiation per d_model class / head-dim class / EXACT / SPLIT); hc = head-dim class 0 (< 16) / 16 / 32 / 64
void gt_seq_launch_pack(const SeqArgs& a, unsigned nblocks, hipStream_t s);
void gt_seq_launch_fwd(const SeqArgs& a, int d_model, int hc, bool split, unsigned nblocks, hipStream_t s);
void gt_seq_launch_bwd(const SeqArgs& a, int d_model, int hc, bool split, unsigned nblocks, hipStream_t s);
