// Weight gradients of the sequence-resident SPLIT path as RIDER workgroups (round 3).
//
// The SPLIT phases of gt_seq.h run 2 x batch workgroups that are LDS-limited to one per CU: at batch 64 half of the chip idles while
// the backward walks its L + 1 launches, and all weight gradients used to follow as one grouped dispatch at the very end (40 of the
// headline step's 249 us).  After backward phase p every operand of layer L-p's FFN / out-proj weight gradients exists (and after
// phase p + 1 its in-proj's), so those products now run INSIDE phase p + 1's launch, as extra workgroups behind the sequence
// workgroups of the same grid ("riders").  A rider is an instance of the same kernel, so it declares the same ~140 KB of LDS: it can
// never share a CU with a sequence workgroup -- it only ever lands on an idle CU (the side-stream form of this overlap, rejected in
// round 2, lost exactly there: co-resident weight-gradient workgroups slowed the LDS-bound phases).
//
// Unit of work: one 32 x 64 tile of dW = dY^T X over a token range, by ONE workgroup:
//   * the 8 waves split the tokens (wave w takes the 8-token slabs w, w + 8, ...), each with its own accumulators and its own
//     double-buffered LDS staging area: no workgroup barrier inside the contraction;
//   * staging: 16-byte line-shaped global loads (a slab row of dY / X is 128 / 256 contiguous bytes), GT_WG_DEPTH slabs in flight in
//     registers; fragments for v_mfma_f32_32x32x2_f32 are conflict-free ds_read_b32 rows (lanes 0-31: token k, lanes 32-63: k + 1);
//   * the 8 partial tiles meet in LDS and are summed in a fixed order: with one owner per tile over ALL tokens the gradient needs no
//     atomics and is bitwise reproducible -- the default now, not an opt-in mode;
//   * bias gradients are the column sums of the dY fragments the MFMAs consume anyway (units of the first column tile).
// The LayerNorm dgamma / dbeta reductions ride as units too (one phase after the partials were written).  What cannot ride -- the
// in-proj of layer 0 and the input layer (their operands appear in the last phase), the 27-wide output layer (packed staging form) --
// runs in seq_tail_kernel on the whole chip, together with the second half of the last phase's tiles and the step-counter bump;
// there a tile's token range is cut in two and the two partial tiles meet in fp32 atomics on the zeroed gradient (order-independent).
// After backward phase p < L everything from encoder layer L - p + 1 to the end of the parameter buffer is final: gradient buckets.
#pragma once
#include "gt_seq_api.h"

#define GT_WG_SLAB 8                      // tokens per staged slab (one wave)
#define GT_WG_TM 32                       // tile rows (columns of dY)
#define GT_WG_TN 64                       // tile columns (columns of X)
#ifndef GT_WG_DEPTH
#define GT_WG_DEPTH 2                     // slabs in flight in registers per wave (2 and 4 time alike; at 4 the rider kernel hits 256 VGPRs and spills)
#endif
#define GT_WG_STAGE (2 * GT_WG_SLAB * (GT_WG_TM + GT_WG_TN))   // floats of staging per wave (two buffers)
#define GT_WG_LDS (8 * GT_WG_TM * GT_WG_TN)                    // floats the unit needs: 8 partial tiles (>= 8 staging areas)

// one weight-gradient problem: C (rows x cols, row stride ldc) (+)= A^T B over the tokens; A (tokens x rows, stride lda) = dY,
// B (tokens x cols, stride ldb) = the layer input; dbias (rows) (+)= column sums of A, or nullptr
struct SeqWgProb { const float* A; const float* B; float* C; float* dbias; int lda, ldb, ldc, rows, cols; };
enum { GT_WG_STORE = 0, GT_WG_ADD = 1, GT_WG_ATOMIC = 2 };

struct SeqWgRegs { float4 a, b0, b1; };
// How an operand's slab (8 tokens x its tile columns) reaches the registers:
//   GT_WGL_LINES  tile wholly inside the operand, 16-byte rows: line-shaped float4 loads (A: 8 lanes per token row, B: 16)
//   GT_WGL_PACKED the operand is narrower than a tile and its rows are contiguous (row stride == width: the 27-wide dlogits, the
//                 16- / 27-wide model input): the slab is ONE contiguous run of 8 x width floats -- float4 loads along the run, the
//                 four elements scattered to their (token, column) slots of the staging tile (columns >= width stay zero)
//   GT_WGL_SCALAR anything else: four guarded 4-byte loads per slot
enum { GT_WGL_LINES = 0, GT_WGL_PACKED = 1, GT_WGL_SCALAR = 2 };
template <int MA, int MB>
__device__ __forceinline__ SeqWgRegs seq_wg_load(const SeqWgProb& p, const int i0, const int j0, const int t0, const int lane, const float* zp) {
  SeqWgRegs r;
  if (MA == GT_WGL_LINES) {
    r.a = *reinterpret_cast<const float4*>(p.A + (size_t)(t0 + (lane >> 3)) * p.lda + i0 + 4 * (lane & 7));
  } else if (MA == GT_WGL_PACKED) {
    r.a = *reinterpret_cast<const float4*>(lane < 2 * p.rows ? p.A + (size_t)t0 * p.lda + 4 * lane : zp);
  } else {
    const int ca = i0 + 4 * (lane & 7);
    const float* pa = p.A + (size_t)(t0 + (lane >> 3)) * p.lda + ca;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = *((ca + e < p.rows) ? pa + e : zp);
    r.a = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (MB == GT_WGL_LINES) {
    const float* pb = p.B + (size_t)(t0 + (lane >> 4)) * p.ldb + j0 + 4 * (lane & 15);
    r.b0 = *reinterpret_cast<const float4*>(pb);
    r.b1 = *reinterpret_cast<const float4*>(pb + (size_t)4 * p.ldb);
  } else if (MB == GT_WGL_PACKED) {
    const float* pb = p.B + (size_t)t0 * p.ldb + 4 * lane;
    r.b0 = *reinterpret_cast<const float4*>(lane < 2 * p.cols ? pb : zp);
    r.b1 = *reinterpret_cast<const float4*>(lane + 64 < 2 * p.cols ? pb + 256 : zp);
  } else {
    const int cb = j0 + 4 * (lane & 15);
    const float* pb = p.B + (size_t)(t0 + (lane >> 4)) * p.ldb + cb;
    float v[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[0][e] = *((cb + e < p.cols) ? pb + e : zp); v[1][e] = *((cb + e < p.cols) ? pb + (size_t)4 * p.ldb + e : zp); }
    r.b0 = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
    r.b1 = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
  }
  return r;
}
// packed operand: staging offsets (token * tile width + column) of the four elements of float4 number f of a slab; -1: beyond the slab
__device__ __forceinline__ void seq_wg_packed_slots(int (&o)[4], const int f, const int width, const int tilew) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int x = 4 * f + e; o[e] = x < GT_WG_SLAB * width ? (x / width) * tilew + x % width : -1; }
}

__device__ __forceinline__ void seq_wg_scatter(float* buf, const int (&o)[4], const float4& v) {
  if (o[0] >= 0) buf[o[0]] = v.x;
  if (o[1] >= 0) buf[o[1]] = v.y;
  if (o[2] >= 0) buf[o[2]] = v.z;
  if (o[3] >= 0) buf[o[3]] = v.w;
}

// One unit: tile (ti, tj) of problem p over tokens [k0, k1) (k0, k1 multiples of 8).  lds: GT_WG_LDS floats; sb: 8 * 64 floats (bias
// partials).  All 512 threads of the workgroup take part; begins and ends with a workgroup barrier.
template <int MA, int MB>
__device__ __forceinline__ void seq_wg_unit(const SeqWgProb& p, const int ti, const int tj, const int k0, const int k1, const int mode,
                                            float* lds, float* sb, const int tid) {
  const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int i0 = GT_WG_TM * ti, j0 = GT_WG_TN * tj;
  const float* const zp = gt_zero_ptr();
  float* const st = lds + wave * GT_WG_STAGE;                    // this wave's staging: [2][8][32] for A, then [2][8][64] for B
  float* const stA = st, * const stB = st + 2 * GT_WG_SLAB * GT_WG_TM;
  const int nslab = (k1 - k0) / GT_WG_SLAB;
  const int nw = nslab > wave ? (nslab - wave + 7) >> 3 : 0;     // slabs of this wave: wave, wave + 8, ...
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  float bsum = 0.f;
  const bool want_bias = p.dbias != nullptr && tj == 0;
  GT_BARRIER();                                                  // (the previous user of lds / sb is done)
  if (nw > 0) {
    // GT_WG_DEPTH slabs in flight, in register sets used in turn (passed and returned BY VALUE: handed to the staging step by reference
    // they came back from the compiler as a scratch array); the trip count is rounded up to a multiple of the depth and the padding
    // slabs re-read the wave's last slab with their dY fragments zeroed: no branch around a load
    auto slab_t0 = [&](int i) { const int ii = i < nw ? i : nw - 1; return k0 + (wave + 8 * ii) * GT_WG_SLAB; };
    SeqWgRegs q[GT_WG_DEPTH];
#pragma unroll
    for (int u = 0; u < GT_WG_DEPTH; ++u) q[u] = seq_wg_load<MA, MB>(p, i0, j0, slab_t0(u), lane, zp);
    const int wa = (lane >> 3) * GT_WG_TM + 4 * (lane & 7);      // this lane's 16-byte slot in an A / B staging buffer
    const int wb = (lane >> 4) * GT_WG_TN + 4 * (lane & 15);
    int oa[4], ob0[4], ob1[4];                                     // packed operands: where this lane's elements go
    if (MA == GT_WGL_PACKED) {
      seq_wg_packed_slots(oa, lane, p.rows, GT_WG_TM);
      for (int e = lane; e < 2 * GT_WG_SLAB * GT_WG_TM; e += 64) stA[e] = 0.f;
    }
    if (MB == GT_WGL_PACKED) {
      seq_wg_packed_slots(ob0, lane, p.cols, GT_WG_TN); seq_wg_packed_slots(ob1, lane + 64, p.cols, GT_WG_TN);
      for (int e = lane; e < 2 * GT_WG_SLAB * GT_WG_TN; e += 64) stB[e] = 0.f;
    }
    GT_WAVE_SYNC();
    auto slab = [&](const SeqWgRegs qq, const int i) -> SeqWgRegs {      // stages qq, returns the same register set reloaded
      float* const bA = stA + (i & 1) * GT_WG_SLAB * GT_WG_TM;
      float* const bB = stB + (i & 1) * GT_WG_SLAB * GT_WG_TN;
      if (MA == GT_WGL_PACKED) {
        seq_wg_scatter(bA, oa, qq.a);
      } else {
        *reinterpret_cast<float4*>(bA + wa) = qq.a;
      }
      if (MB == GT_WGL_PACKED) {
        seq_wg_scatter(bB, ob0, qq.b0);
        seq_wg_scatter(bB, ob1, qq.b1);
      } else {
        *reinterpret_cast<float4*>(bB + wb) = qq.b0;
        *reinterpret_cast<float4*>(bB + wb + 4 * GT_WG_TN) = qq.b1;
      }
      GT_WAVE_SYNC();                                              // the slab is written by all 64 lanes before any lane reads fragments
      const SeqWgRegs nq = seq_wg_load<MA, MB>(p, i0, j0, slab_t0(i + GT_WG_DEPTH), lane, zp);
      const float live = i < nw ? 1.0f : 0.0f;
#pragma unroll
      for (int kk = 0; kk < GT_WG_SLAB / 2; ++kk) {
        const float av = bA[(2 * kk + h) * GT_WG_TM + r] * live;
        const float b0 = bB[(2 * kk + h) * GT_WG_TN + r], b1 = bB[(2 * kk + h) * GT_WG_TN + 32 + r];
        acc0 = GT_MFMA32(av, b0, acc0);
        acc1 = GT_MFMA32(av, b1, acc1);
        bsum += av;
      }
      GT_WAVE_SYNC();                                              // (emulator: nobody overwrites this buffer's twin while a lane still reads)
      return nq;
    };
    for (int i = 0; i < nw; i += GT_WG_DEPTH) {
#pragma unroll
      for (int u = 0; u < GT_WG_DEPTH; ++u) q[u] = slab(q[u], i + u);
    }
  }
  GT_BARRIER();                                                  // every wave is done with its staging area: the partial tiles go over them
  {
    float* const pt = lds + wave * (GT_WG_TM * GT_WG_TN);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      pt[row * GT_WG_TN + r] = acc0[e];
      pt[row * GT_WG_TN + 32 + r] = acc1[e];
    }
    sb[wave * 64 + lane] = bsum;
  }
  GT_BARRIER();
  // sum of the 8 partial tiles, wave 0's first: thread (row = tid >> 4, seg = tid & 15) -> columns seg + 16 e (64-byte row segments
  // per 16 lanes: the shape plain stores and float atomics both like)
  {
    const int row = tid >> 4, seg = tid & 15;
    float s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] = lds[row * GT_WG_TN + seg + 16 * e];
#pragma unroll
      for (int w = 1; w < 8; ++w) s[e] += lds[w * (GT_WG_TM * GT_WG_TN) + row * GT_WG_TN + seg + 16 * e];
    }
    if (i0 + row < p.rows) {
      float* c = p.C + (size_t)(i0 + row) * p.ldc + j0 + seg;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (j0 + seg + 16 * e < p.cols) {
          if (mode == GT_WG_ATOMIC) atomicAdd(c + 16 * e, s[e]);
          else if (mode == GT_WG_ADD) c[16 * e] += s[e];
          else c[16 * e] = s[e];
        }
      }
    }
    if (want_bias && tid < GT_WG_TM && i0 + tid < p.rows) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += sb[w * 64 + tid] + sb[w * 64 + 32 + tid];
      if (mode == GT_WG_ATOMIC) atomicAdd(p.dbias + i0 + tid, t);
      else if (mode == GT_WG_ADD) p.dbias[i0 + tid] += t;
      else p.dbias[i0 + tid] = t;
    }
  }
}
// tiles of a problem
__device__ __forceinline__ int seq_wg_tiles(const int rows, const int cols) { return ((rows + GT_WG_TM - 1) / GT_WG_TM) * ((cols + GT_WG_TN - 1) / GT_WG_TN); }
template <bool TAIL>
__device__ __forceinline__ void seq_wg_run(const SeqWgProb& p, const int tile, const int k0, const int k1, const int mode, float* lds, float* sb,
                                           const int tid) {
  const int ntj = (p.cols + GT_WG_TN - 1) / GT_WG_TN, ti = tile / ntj, tj = tile % ntj;
  // wave-uniform choice of the staging form per operand (see GT_WGL_*)
  const bool al_a = (reinterpret_cast<uintptr_t>(p.A) & 15) == 0, al_b = (reinterpret_cast<uintptr_t>(p.B) & 15) == 0;
  const bool lines_a = GT_WG_TM * (ti + 1) <= p.rows && (p.lda & 3) == 0 && al_a;
  const bool lines_b = GT_WG_TN * (tj + 1) <= p.cols && (p.ldb & 3) == 0 && al_b;
  const bool packed_a = p.rows <= GT_WG_TM && p.lda == p.rows && al_a;
  const bool packed_b = p.cols <= GT_WG_TN && p.ldb == p.cols && al_b;
  if (lines_a && lines_b) seq_wg_unit<GT_WGL_LINES, GT_WGL_LINES>(p, ti, tj, k0, k1, mode, lds, sb, tid);
  else if (TAIL && packed_a && lines_b) seq_wg_unit<GT_WGL_PACKED, GT_WGL_LINES>(p, ti, tj, k0, k1, mode, lds, sb, tid);    // the output layer
  else if (TAIL && lines_a && packed_b) seq_wg_unit<GT_WGL_LINES, GT_WGL_PACKED>(p, ti, tj, k0, k1, mode, lds, sb, tid);    // the input layer
  else seq_wg_unit<GT_WGL_SCALAR, GT_WGL_SCALAR>(p, ti, tj, k0, k1, mode, lds, sb, tid);
}

// ---- the problems of the encoder (operands: the buffers the sequence kernels save; destinations: a.grd at the parameter offsets)
enum { GT_WGP_OUT = 0, GT_WGP_W2 = 1, GT_WGP_W1 = 2, GT_WGP_WO = 3, GT_WGP_WIN = 4, GT_WGP_IN = 5, GT_WGP_LN = 6 };
__device__ __forceinline__ SeqWgProb seq_wg_prob(const SeqArgs& a, const int kind, const int l) {
  const int d = a.d, F = a.F;
  const float* ws = a.ws;
  const float* wl = ws + (int64_t)l * a.wstride;
  const float* tl = ws + (int64_t)l * a.tstride;
  float* g = a.grd + (int64_t)l * a.pstride;
  const bool drop = a.st != nullptr && a.thr != 0u;
  SeqWgProb p;
  switch (kind) {
    case GT_WGP_OUT: p = SeqWgProb{ws + a.dlogits, ws + a.memory, a.grd + a.out_w, a.grd + a.out_b, GT_TGT, d, d, GT_TGT, d}; break;
    case GT_WGP_W2:  p = SeqWgProb{tl + (drop ? a.t0.dzAm : a.t0.dzA), wl + a.w0.hact, g + a.p0.w2, g + a.p0.b2, d, F, F, d, F}; break;
    case GT_WGP_W1:  p = SeqWgProb{tl + a.t0.dhid, wl + a.w0.x1, g + a.p0.w1, g + a.p0.b1, F, d, d, F, d}; break;
    case GT_WGP_WO:  p = SeqWgProb{tl + (drop ? a.t0.dzBm : a.t0.dzB), wl + a.w0.ctx, g + a.p0.out_w, g + a.p0.out_b, d, d, d, d, d}; break;
    case GT_WGP_WIN: p = SeqWgProb{tl + a.t0.dqkv, l == 0 ? ws + a.x0 : ws + (int64_t)(l - 1) * a.wstride + a.w0.xout, g + a.p0.in_w, g + a.p0.in_b,
                                   3 * d, d, d, 3 * d, d}; break;
    default:         p = SeqWgProb{ws + a.da0, a.xin, a.grd + a.in_w, a.grd + a.in_b, d, a.S, a.S, d, a.S}; break;
  }
  return p;
}
// LayerNorm dgamma / dbeta: job j (order of the sequence kernels' partial blocks: 0 = final norm, 1 + 2 k = norm2 of layer L-1-k,
// 2 + 2 k = its norm1), 64 columns of [dgamma | dbeta] per workgroup (column block cb): thread (column, row group of 8); a group walks
// its partial rows 8 loads at a time (a one-load-per-trip walk is a chain of L2 round trips: 32 of them took the first tail to 37 us),
// the groups meet in LDS, group 0 first -- a fixed order.  Begins and ends with a workgroup barrier.
__device__ __forceinline__ void seq_wg_ln_job(const SeqArgs& a, const int j, const int cb, float* lds, const int tid) {
  const int d = a.d, nwg = a.ln_nwg, ncol = 2 * d;
  int64_t goff;
  if (j == 0) goff = a.encn_w;
  else { const int k = (j - 1) >> 1, l = a.L - 1 - k; goff = (int64_t)l * a.pstride + (((j - 1) & 1) ? a.p0.n1w : a.p0.n2w); }
  const float* part = a.ws + a.ln_part + (int64_t)j * a.ln_part_stride;        // [nwg][2][d]
  float* dst = a.grd + goff;                                                      // dgamma; dbeta follows at the next 64-float boundary
  const int64_t bo = (d + 63) / 64 * 64;
  const float* const zp = gt_zero_ptr();
  const int c = cb * 64 + (tid & 63), grp = tid >> 6;
  const bool ok = c < ncol;
  float acc = 0.f;
  for (int g0 = grp; g0 < nwg; g0 += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int g = g0 + 8 * u; v[u] = *((ok && g < nwg) ? part + (size_t)g * ncol + c : zp); }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  GT_BARRIER();
  lds[tid] = acc;
  GT_BARRIER();
  if (grp == 0 && ok) {
    float t = lds[tid];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += lds[tid + 64 * q];
    float* o = c < d ? dst + c : dst + bo + (c - d);
    if (a.wg_accumulate) *o += t; else *o = t;
  }
}
// The work whose operands are complete when backward phase `phase` STARTS and not earlier (phase L + 1: the tail), in launch
// order: f(kind, layer or LayerNorm job) for each until f returns true.  (No arrays: everything stays in scalar registers.)
//   phase 0      (riders: nothing) the output layer -- its operands exist before the backward starts, but its 27-wide dY wants the
//                packed staging form, which only the tail kernel carries (in the rider kernel it cost 880 B/lane of scratch): list 0
//                is run by a tail-kernel launch of its own, and only when the backward is cut in two for the data-parallel overlap
//   phase p >= 1 in-proj of layer L-p+1 (p >= 2), FFN2 / FFN1 / out-proj of layer L-p, the LayerNorm jobs phase p-1 filled
//   tail         in-proj of layer 0, output layer (unless list 0 ran), input layer
// so after phase p everything from encoder layer L-p+1 to the end of the parameter buffer is final: gradient buckets for the
// data-parallel all-reduce (gt_grad_buckets).
template <typename F>
__device__ __forceinline__ void seq_wg_phase_list(const SeqArgs& a, const int phase, F f) {
  const int L = a.L;
  if (phase == 0) { f(GT_WGP_OUT, 0); return; }
  if (phase >= 2) { if (f(GT_WGP_WIN, L - phase + 1)) return; }
  if (phase <= L) {
    const int l = L - phase;
    if (f(GT_WGP_W2, l)) return;
    if (f(GT_WGP_W1, l)) return;
    if (f(GT_WGP_WO, l)) return;
    if (phase == 1) { if (f(GT_WGP_LN, 0)) return; }
    if (f(GT_WGP_LN, 2 * phase - 1)) return;
    f(GT_WGP_LN, 2 * phase);
  } else {
    if (!a.out_early) { if (f(GT_WGP_OUT, 0)) return; }
    f(GT_WGP_IN, 0);
  }
}
__device__ __forceinline__ int seq_wg_kind_tiles(const SeqArgs& a, const int kind) {
  const int d = a.d, F = a.F;
  switch (kind) {
    case GT_WGP_OUT: return seq_wg_tiles(GT_TGT, d);
    case GT_WGP_W2:  return seq_wg_tiles(d, F);
    case GT_WGP_W1:  return seq_wg_tiles(F, d);
    case GT_WGP_WO:  return seq_wg_tiles(d, d);
    case GT_WGP_WIN: return seq_wg_tiles(3 * d, d);
    case GT_WGP_LN:  return (2 * d + 63) / 64;
    default:         return seq_wg_tiles(d, a.S);
  }
}
// units of a phase's list: matrix tiles x ksplit token chunks (+ the LayerNorm column blocks, never split, when with_ln)
__device__ __forceinline__ int seq_wg_phase_units(const SeqArgs& a, const int phase, const int ksplit, const bool with_ln) {
  int n = 0;
  seq_wg_phase_list(a, phase, [&](int kind, int) {
    if (kind == GT_WGP_LN) { if (with_ln) n += seq_wg_kind_tiles(a, kind); }
    else n += seq_wg_kind_tiles(a, kind) * ksplit;
    return false;
  });
  return n;
}
// unit u of the phase's list over the tokens [klo, khi), split into ksplit chunks (units = tiles x chunks, chunk-major inside a tile);
// false when u is beyond the list
template <bool TAIL>
__device__ __forceinline__ bool seq_wg_phase_unit(const SeqArgs& a, const int phase, int u, const int klo, const int khi, const int ksplit,
                                                  const int mode, const bool with_ln, float* lds, float* sb, const int tid) {
  bool done = false;
  seq_wg_phase_list(a, phase, [&](int kind, int layer) {
    if (kind == GT_WGP_LN) {
      if (!with_ln) return false;
      const int ncb = seq_wg_kind_tiles(a, kind);
      if (u >= ncb) { u -= ncb; return false; }
      seq_wg_ln_job(a, layer, u, lds, tid);
      done = true;
      return true;
    }
    const int nt = seq_wg_kind_tiles(a, kind) * ksplit;
    if (u >= nt) { u -= nt; return false; }
    const SeqWgProb p = seq_wg_prob(a, kind, layer);
    const int tile = u / ksplit, c = u % ksplit;
    const int per = (((khi - klo) / GT_WG_SLAB + ksplit - 1) / ksplit) * GT_WG_SLAB;        // tokens per chunk (multiple of 8)
    const int k0 = klo + c * per < khi ? klo + c * per : khi, k1 = k0 + per < khi ? k0 + per : khi;
    seq_wg_run<TAIL>(p, tile, k0, k1, mode, lds, sb, tid);
    done = true;
    return true;
  });
  return done;
}
// rider workgroup r of R in backward phase `phase`: units r, r + R, ... of the phase's list.  Every phase but the last covers all
// tokens; the LAST phase (its sequence work is short: attention backward + in-proj dgrad of layer 0) covers [0, a.ride_last_k) and
// leaves the rest of each tile to the tail launch, which ADDS behind it -- a split across two launches needs no atomics.
__device__ __forceinline__ void seq_wg_riders(const SeqArgs& a, const int phase, const int r, const int R, float* lds, float* sb, const int tid) {
  const int mode = a.wg_accumulate ? GT_WG_ADD : GT_WG_STORE;
  const int khi = phase == a.L ? a.ride_last_k : a.B * 32;
  for (int u = r; seq_wg_phase_unit<false>(a, phase, u, 0, khi, 1, mode, true, lds, sb, tid); u += R) { }
}

#ifdef GT_SEQ_TU_BWD
// ---- the tail, in block order: (1) the rest of the last phase's tiles (tokens [a.ride_last_k, M), added behind the riders' part);
// (2) what could not ride at all -- layer 0's in-proj and the input layer -- token range split a.tail_ksplit ways (two partial tiles
// meeting in fp32 atomics on a zeroed gradient are still order-independent; more are not: gt_set_deterministic keeps it at <= 2);
// and the step-counter bump of the fused train step.
__global__ __launch_bounds__(GT_SEQ_NT_WG) void seq_tail_kernel(SeqArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[GT_WG_LDS];
  __shared__ float sb[8 * 64];
  const int tid = threadIdx.x;
  if (a.bump != nullptr && blockIdx.x == 0 && tid == 0) gt_bump_counters(a.bump, a.xchg >= 0 ? reinterpret_cast<unsigned*>(a.ws + a.xchg) : nullptr, nullptr);
  const int M = a.B * 32, ks = a.tail_ksplit;
  int blk = blockIdx.x;
  if (a.tail_phase <= a.L) {       // (a debug launch names one phase's list: its matrix tiles alone, over all tokens)
    seq_wg_phase_unit<true>(a, a.tail_phase, blk, 0, M, ks, ks > 1 ? GT_WG_ATOMIC : GT_WG_ADD, false, lds, sb, tid);
    return;
  }
  const int nrest = a.ride_last_k < M ? seq_wg_phase_units(a, a.L, 1, false) : 0;
  if (blk < nrest) { seq_wg_phase_unit<true>(a, a.L, blk, a.ride_last_k, M, 1, GT_WG_ADD, false, lds, sb, tid); return; }
  blk -= nrest;
  seq_wg_phase_unit<true>(a, a.L + 1, blk, 0, M, ks, ks > 1 ? GT_WG_ATOMIC : (a.wg_accumulate ? GT_WG_ADD : GT_WG_STORE), true, lds, sb, tid);
}
#endif
