// Sequence-resident kernels for small models (SURVEY 7 "regime i": the reference's shipped YAMLs have d_model 32).
//
// One workgroup owns ONE sequence -- 32 tokens, the whole attention window -- and walks the entire encoder with it: every
// stage of every layer (in-proj, attention, out-proj, LayerNorm, FFN, LayerNorm ... output heads; and the whole chain back)
// runs in the same launch with workgroup barriers between stages.  No stage ever needs another sequence, so there is no
// grid-level synchronisation: the forward of the model is ONE launch and so is its backward (weight gradients still leave as
// the grouped dispatch of gt_gemm.h: they contract over all sequences).  At d_model 32 the step's 33-49 launches of ~5-8 us were
// nothing but launch / hand-over latency (0.6-1.3 % of the MFMA peak); the tile of a sequence is 4 KB and stays in LDS.
//
//   activations of the sequence   LDS: x, x1, ctx tiles [32][DP + 8], the qkv tile, the FFN tile [32][F + 8] (A operands, k contiguous)
//   weights                       read once per workgroup, straight from L2 into MFMA B fragments (no LDS staging: every wave
//                                 owns its output columns, so no weight element is needed twice)
//   saved for backward / tests    written to the same workspace buffers the one-kernel-per-op path uses (gt_ws_find names)
//   attention                     transposed-score MFMA bodies (the scheme of gt_attn.h) on LDS operands, four heads at a time
//
// What bounds these kernels (in-kernel stamps, profiles/r02_seq_stamps.txt): a workgroup is a chain of ~8 dependent stages per
// layer and each stage is ONE wave's instruction stream per SIMD -- a wave64 VALU instruction occupies its SIMD for 4 cycles
// (8 with the SIMD's second wave), a quarter-rate v_mul_lo_u32 for 16, a dependent ds_bpermute / LDS access ~100.  So the rules
// here are instruction-count rules: 32-bit element offsets from wave-uniform bases (no 64-bit address arithmetic per load),
// 16-byte LDS / global accesses, wave-uniform branches instead of per-lane zero-page selects, every elementwise pass spread
// over all 512 threads (16 lanes per token row), row reductions by DPP (no LDS round trip), the dropout state fetched once.
//
// Supported: encoder-only, fp32 operands, d_model % 16 == 0 and <= 64, dim_feedforward % 16 == 0 and <= 512, src_dim <= 32,
// head_dim 16 / 32 / 64 or < 16 (seq_supported in groove_hip.hip).
#pragma once
#include "gt_attn.h"
#include "gt_gemm.h"

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
};
// In-kernel stamps (diagnostic build only; cdna_hip_programming.md 7): workgroup 0, thread 0 records the shader clock at stage
// boundaries into a buffer nothing else reads.  tools/seq_stamps.py prints the per-stage cycle counts.
#ifdef GT_SEQ_STAMPS
#define GT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(a.ws + a.stamps)[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GT_STAMP(i) do { } while (0)
#endif

// Stage hand-overs inside these kernels go through LDS only (the global stores are copies for LATER kernels: saved activations,
// weight-gradient operands), so the stage barrier is GT_BARRIER(): it waits for this wave's LDS traffic, not for its
// outstanding global stores.
#define GT_SEQ_WAVES 8
#define GT_SEQ_NT (GT_SEQ_WAVES * 64)

// ---- dropout: the step state is read ONCE per kernel (three scalars); keys and multipliers are the ones of gt_common.h ------------
struct SeqDropK { uint32_t thr; float scale; uint32_t s_lo, s_hi; };
__device__ __forceinline__ SeqDropK seq_dropk(const SeqArgs& a) {
  SeqDropK k;
  k.thr = (a.st != nullptr) ? a.thr : 0u; k.scale = a.dscale; k.s_lo = 0u; k.s_hi = 0u;
  if (k.thr) { k.s_lo = a.st->seed_lo ^ gt_fmix32(a.st->step); k.s_hi = a.st->seed_hi; }
  return k;
}
__device__ __forceinline__ uint32_t seq_key(const SeqDropK& k, const int site) {       // == gt_drop_key
  if (!k.thr) return 0u;
  uint32_t x = k.s_lo ^ ((uint32_t)site * 0x9E3779B9u);
  x = gt_fmix32(x) ^ k.s_hi;
  return gt_fmix32(x + 0x7F4A7C15u);
}
__device__ __forceinline__ float seq_dmul(const SeqDropK& k, const uint32_t key, const uint32_t idx) {   // == gt_drop_mul
  if (!k.thr) return 1.0f;
  return ((gt_fmix32((idx * 0x9E3779B1u) ^ key) >> 8) >= k.thr) ? k.scale : 0.0f;
}

// ---- sums over the 16 lanes of a token row (lanes 16 r .. 16 r + 15 of a wave = one DPP row): quad butterfly, then the mirrored
// half-row and row -- the same operand pairs as an xor-1/2/4/8 butterfly (bit-identical to it), without the LDS crossbar.
__device__ __forceinline__ float seq_row16_sum(float v) {
#ifdef GT_EMU
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
#else
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
#endif
  return v;
}

// ================================================================================================================ matmul primitives
// C (32 x N) = A (32 x K, an LDS tile, k contiguous) * B, B from global memory straight into MFMA fragments.
//   BKM = false: B(k, n) = W[n * ldw + k] (Linear forward);  BKM = true: B(k, n) = W[k * ldw + n] (dgrad).
// A tile of 16 output columns: lane (l16, lg) holds rows l16 (acc0) and 16 + l16 (acc1), columns n0 + 4 lg + 0..3.
// Every global load of a stage is issued before the first MFMA (one memory round trip per stage).

// ---- edge version (the 27-wide input / output layers, once per kernel): any K <= 64, any N; masked through the zero page
template <bool BKM, bool VEC>
__device__ __forceinline__ void seq_ldb(float (&b)[4], const float* __restrict__ W, const int ldw, const int n, const int N, const int k,
                                        const int K, const float* zp) {
  if (!BKM && VEC) {               // K % 4 == 0, 16-byte rows: a float4 is in range or out as a whole
    const float4 v = *reinterpret_cast<const float4*>((n < N && k < K) ? W + (size_t)n * ldw + k : zp);
    b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
  } else if (!BKM) {               // rows of any length (the 27-wide symbolic input)
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *((n < N && k + j < K) ? W + (size_t)n * ldw + k + j : zp);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *((n < N && k + j < K) ? W + (size_t)(k + j) * ldw + n : zp);
  }
}
__device__ __forceinline__ void seq_mma(f32x4& acc0, f32x4& acc1, const float (&b)[4], const float* sA, const int lda, const int l16, const int k) {
  const float4 a0 = *reinterpret_cast<const float4*>(sA + l16 * lda + k);
  const float4 a1 = *reinterpret_cast<const float4*>(sA + (16 + l16) * lda + k);
  acc0 = GT_MFMA16(b[0], a0.x, acc0); acc1 = GT_MFMA16(b[0], a1.x, acc1);
  acc0 = GT_MFMA16(b[1], a0.y, acc0); acc1 = GT_MFMA16(b[1], a1.y, acc1);
  acc0 = GT_MFMA16(b[2], a0.z, acc0); acc1 = GT_MFMA16(b[2], a1.z, acc1);
  acc0 = GT_MFMA16(b[3], a0.w, acc0); acc1 = GT_MFMA16(b[3], a1.w, acc1);
}
// wave w owns tile w (N <= 128).  epi(n0, acc0, acc1, bias4): bias4 = bias[n0 + 4 lg + 0..3] (zeros without a bias).
template <bool BKM, bool VEC, typename Epi>
__device__ __forceinline__ void seq_mm_edge(const float* sA, const int lda, const int K, const float* __restrict__ W, const int ldw, const int N,
                                            const float* __restrict__ bias, const int wave, const int lane, const float* zp, Epi epi) {
  const int l16 = lane & 15, lg = lane >> 4;
  const int n0 = wave * 16;
  if (n0 >= N) return;                                   // wave-uniform
  float b[4][4], bi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) seq_ldb<BKM, VEC>(b[u], W, ldw, n0 + l16, N, 16 * u + 4 * lg, K, zp);
#pragma unroll
  for (int r = 0; r < 4; ++r) bi[r] = *((bias != nullptr && n0 + 4 * lg + r < N) ? bias + n0 + 4 * lg + r : zp);
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) { if (16 * u < K) seq_mma(acc0, acc1, b[u], sA, lda, l16, 16 * u + 4 * lg); }
  epi(n0, acc0, acc1, bi);
}

// ---- B fragment of one 16-column tile, K % 16 == 0 and <= 64, N % 16 == 0: 32-bit offsets from the wave-uniform W
template <bool BKM>
__device__ __forceinline__ void seq_ldb_tile(float4 (&b)[4], const float* __restrict__ W, const int ldw, const int n0, const int K, const int l16,
                                             const int lg) {
  if (!BKM) {
    const float* wp = W + (unsigned)((n0 + l16) * ldw + 4 * lg);
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (16 * u < K) b[u] = *reinterpret_cast<const float4*>(wp + 16 * u); }
  } else {
    const float* wp = W + (unsigned)(4 * lg * ldw + n0 + l16);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (16 * u < K) {
        b[u].x = wp[(unsigned)((16 * u + 0) * ldw)]; b[u].y = wp[(unsigned)((16 * u + 1) * ldw)];
        b[u].z = wp[(unsigned)((16 * u + 2) * ldw)]; b[u].w = wp[(unsigned)((16 * u + 3) * ldw)];
      }
    }
  }
}
__device__ __forceinline__ void seq_mma4(f32x4& acc, const float4& b, const float4& a) {
  acc = GT_MFMA16(b.x, a.x, acc); acc = GT_MFMA16(b.y, a.y, acc); acc = GT_MFMA16(b.z, a.z, acc); acc = GT_MFMA16(b.w, a.w, acc);
}
// Short contraction (K % 16 == 0, <= 64), N % 16 == 0: wave w owns tiles w, w + 8, ... (at most MAXT of them).  The A fragments
// (both 16-row halves) are read from LDS once and reused by all tiles of the wave.  epi(n0, acc0, acc1, bias float4).
template <bool BKM, int MAXT, typename Epi>
__device__ __forceinline__ void seq_mm_tiles(const float* sA, const int lda, const int K, const float* __restrict__ W, const int ldw, const int N,
                                             const float* __restrict__ bias, const int wave, const int lane, Epi epi) {
  const int l16 = lane & 15, lg = lane >> 4, ntile = N >> 4;
  if (wave >= ntile) return;                             // wave-uniform
  float4 b[MAXT][4], bi[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + i * GT_SEQ_WAVES;
    if (t < ntile) {
      seq_ldb_tile<BKM>(b[i], W, ldw, 16 * t, K, l16, lg);
      bi[i] = bias != nullptr ? *reinterpret_cast<const float4*>(bias + 16 * t + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 a0[4], a1[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (16 * u < K) {
      a0[u] = *reinterpret_cast<const float4*>(sA + l16 * lda + 16 * u + 4 * lg);
      a1[u] = *reinterpret_cast<const float4*>(sA + (16 + l16) * lda + 16 * u + 4 * lg);
    }
  }
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + i * GT_SEQ_WAVES;
    if (t < ntile) {
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (16 * u < K) {
          acc0 = GT_MFMA16(b[i][u].x, a0[u].x, acc0); acc1 = GT_MFMA16(b[i][u].x, a1[u].x, acc1);
          acc0 = GT_MFMA16(b[i][u].y, a0[u].y, acc0); acc1 = GT_MFMA16(b[i][u].y, a1[u].y, acc1);
          acc0 = GT_MFMA16(b[i][u].z, a0[u].z, acc0); acc1 = GT_MFMA16(b[i][u].z, a1[u].z, acc1);
          acc0 = GT_MFMA16(b[i][u].w, a0[u].w, acc0); acc1 = GT_MFMA16(b[i][u].w, a1[u].w, acc1);
        }
      }
      epi(16 * t, acc0, acc1, bi[i]);
    }
  }
}
// Square projection (N = K = d <= 64): 2 * (d / 16) <= 8 units of (column tile, 16-row half), one per wave -- the raw 16 x 16
// results go to an LDS tile [32][srs]; bias / dropout / residual belong to the LayerNorm pass that reads it (all 512 threads).
template <bool BKM>
__device__ __forceinline__ void seq_mm_square(const float* sA, const int lda, const int d, const float* __restrict__ W, float* sOut, const int srs,
                                              const int wave, const int lane) {
  const int l16 = lane & 15, lg = lane >> 4, t = wave >> 1, half = wave & 1;
  if (16 * t >= d) return;
  float4 b[4], av[4];
  seq_ldb_tile<BKM>(b, W, d, 16 * t, d, l16, lg);
#pragma unroll
  for (int u = 0; u < 4; ++u) { if (16 * u < d) av[u] = *reinterpret_cast<const float4*>(sA + (16 * half + l16) * lda + 16 * u + 4 * lg); }
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) { if (16 * u < d) seq_mma4(acc, b[u], av[u]); }
  *reinterpret_cast<float4*>(sOut + (16 * half + l16) * srs + 16 * t + 4 * lg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}
// Long contraction (K % 16 == 0, up to 512) into few columns (N <= 64: NT = N / 16 <= 4 tiles): the 8 waves split K as well --
// wave w takes tile w % NT and k-part w / NT of KS = 8 / NT parts -- and leave partial tiles in sR[part][32][srs]; the pass that
// reads them (seq_parts_sum, inside the following LayerNorm pass) sums the parts in a fixed order, part 0 first.
__device__ __forceinline__ int seq_splitk_parts(const int N) { return GT_SEQ_WAVES / ((N + 15) >> 4); }
template <bool BKM>
__device__ __forceinline__ void seq_mm_splitk(const float* sA, const int lda, const int K, const float* __restrict__ W, const int ldw, const int N,
                                              float* sR, const int srs, const int wave, const int lane) {
  const int l16 = lane & 15, lg = lane >> 4;
  const int NT = N >> 4, KS = GT_SEQ_WAVES / NT;                  // NT in {1, 2, 4} -> KS in {8, 4, 2}; NT = 3 -> KS = 2 (two waves idle)
  const int t = wave % NT, part = wave / NT;
  if (part >= KS) return;
  const int nks = K >> 4, per = (nks + KS - 1) / KS, ks0 = part * per, ks1 = (ks0 + per < nks) ? ks0 + per : nks;
  const int n0 = t * 16;
  const float* wp = BKM ? W + (unsigned)(4 * lg * ldw + n0 + l16) : W + (unsigned)((n0 + l16) * ldw + 4 * lg);
  const float* ap = sA + l16 * lda + 4 * lg;
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c0 = ks0; c0 < ks1; c0 += 8) {                          // at most 8 k-steps (128 k) per round trip; K 512 / KS 2 -> two
    float4 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (c0 + u < ks1) {                                          // wave-uniform
        const int k = 16 * (c0 + u);
        if (!BKM) b[u] = *reinterpret_cast<const float4*>(wp + k);
        else {
          b[u].x = wp[(unsigned)((k + 0) * ldw)]; b[u].y = wp[(unsigned)((k + 1) * ldw)];
          b[u].z = wp[(unsigned)((k + 2) * ldw)]; b[u].w = wp[(unsigned)((k + 3) * ldw)];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (c0 + u < ks1) {
        const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * (c0 + u)), a1 = *reinterpret_cast<const float4*>(ap + 16 * lda + 16 * (c0 + u));
        acc0 = GT_MFMA16(b[u].x, a0.x, acc0); acc1 = GT_MFMA16(b[u].x, a1.x, acc1);
        acc0 = GT_MFMA16(b[u].y, a0.y, acc0); acc1 = GT_MFMA16(b[u].y, a1.y, acc1);
        acc0 = GT_MFMA16(b[u].z, a0.z, acc0); acc1 = GT_MFMA16(b[u].z, a1.z, acc1);
        acc0 = GT_MFMA16(b[u].w, a0.w, acc0); acc1 = GT_MFMA16(b[u].w, a1.w, acc1);
      }
    }
  }
  float* r = sR + part * 32 * srs + n0 + 4 * lg;
  *reinterpret_cast<float4*>(r + l16 * srs) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
  *reinterpret_cast<float4*>(r + (16 + l16) * srs) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
}

// ================================================================================================================ row passes
// Elementwise / LayerNorm passes over a [32][d] tile use ALL 512 threads: thread (row = tid >> 4, seg = tid & 15) owns the
// CW = DP / 16 columns seg * CW ..; the 16 lanes of a row are one DPP row.  d % 16 == 0, so a thread's columns are in range
// (c0 < d) or out as a whole.
template <int CW> struct SeqVec;
template <> struct SeqVec<2> {
  static __device__ __forceinline__ void ld(float (&v)[2], const float* p) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <> struct SeqVec<4> {
  static __device__ __forceinline__ void ld(float (&v)[4], const float* p) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
// sum of the split-K partial tiles of this thread's columns, part 0 first
template <int CW>
__device__ __forceinline__ void seq_parts_sum(float (&v)[CW], const float* sR, const int srs, const int parts, const int row, const int c0) {
  SeqVec<CW>::ld(v, sR + row * srs + c0);
  for (int p = 1; p < parts; ++p) {
    float u[CW];
    SeqVec<CW>::ld(u, sR + (p * 32 + row) * srs + c0);
#pragma unroll
    for (int e = 0; e < CW; ++e) v[e] += u[e];
  }
}

// LayerNorm forward: z (this thread's CW values, from zfun(row, c0, z)) -> y = LN(z) gamma + beta -> the LDS tile sY and the
// global y / xhat / rstd rows of this sequence (gy, gxhat, grstd: wave-uniform bases of the sequence's first row)
template <int DP, typename ZFun>
__device__ __forceinline__ void seq_ln_fwd(ZFun zfun, float* sY, const int str, const int d, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float* gy, float* gxhat, float* grstd, const int tid) {
  constexpr int CW = DP / 16;
  const int row = tid >> 4, seg = tid & 15, c0 = seg * CW;
  const bool ok = c0 < d;
  float z[CW], ga[CW], be[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) { z[e] = 0.f; ga[e] = 0.f; be[e] = 0.f; }
  if (ok) { SeqVec<CW>::ld(ga, gamma + c0); SeqVec<CW>::ld(be, beta + c0); zfun(row, c0, z); }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < CW; ++e) s += z[e];
  const float invd = 1.0f / (float)d, mean = seq_row16_sum(s) * invd;
  float q = 0.f;
  if (ok) {
#pragma unroll
    for (int e = 0; e < CW; ++e) { const float t = z[e] - mean; q += t * t; }
  }
  const float rs = 1.0f / sqrtf(seq_row16_sum(q) * invd + GT_LN_EPS);
  if (ok) {
    float xh[CW], y[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) { xh[e] = (z[e] - mean) * rs; y[e] = xh[e] * ga[e] + be[e]; }
    const unsigned o = (unsigned)(row * d + c0);
    SeqVec<CW>::st(sY + row * str + c0, y);
    SeqVec<CW>::st(gy + o, y);
    SeqVec<CW>::st(gxhat + o, xh);
  }
  if (seg == 0) grstd[row] = rs;
}

// LayerNorm backward: g (from gfun) -> dz = LNbwd(g) -> sDz (LDS, unmasked: the residual gradient), dz * dropout mask -> sDzm
// (LDS: the next dgrad's A operand), both to global when gdz / gdzm are given (weight-gradient operands); the per-wave column
// sums of g xhat / g (4 rows each) -> sP[wave][2][DP]; seq_ln_part sums them over the waves after the stage barrier.
template <int DP, typename GFun>
__device__ __forceinline__ void seq_ln_bwd(GFun gfun, float* sDz, float* sDzm, const int str, const int d, const float* __restrict__ gxhat,
                                           const float* __restrict__ grstd, const float* __restrict__ gamma, const SeqDropK& dk, const uint32_t key,
                                           const uint32_t idx0, float* gdz, float* gdzm, float* sP, const int tid) {
  constexpr int CW = DP / 16;
  const int row = tid >> 4, seg = tid & 15, c0 = seg * CW, lane = tid & 63, wave = tid >> 6;
  const bool ok = c0 < d;
  const unsigned o = (unsigned)(row * d + c0);
  float g[CW], xh[CW], ga[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) { g[e] = 0.f; xh[e] = 0.f; ga[e] = 0.f; }
  const float rs = grstd[row];
  if (ok) { SeqVec<CW>::ld(xh, gxhat + o); SeqVec<CW>::ld(ga, gamma + c0); gfun(row, c0, g); }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < CW; ++e) { const float gd = g[e] * ga[e]; s1 += gd; s2 += gd * xh[e]; }
  const float invd = 1.0f / (float)d, m1 = seq_row16_sum(s1) * invd, m2 = seq_row16_sum(s2) * invd;
  float cg[CW], cb[CW];                                          // column sums over the 4 rows of this wave
#pragma unroll
  for (int e = 0; e < CW; ++e) { cg[e] = g[e] * xh[e]; cb[e] = g[e]; }
#pragma unroll
  for (int e = 0; e < CW; ++e) { cg[e] += __shfl_xor(cg[e], 16); cb[e] += __shfl_xor(cb[e], 16); }
#pragma unroll
  for (int e = 0; e < CW; ++e) { cg[e] += __shfl_xor(cg[e], 32); cb[e] += __shfl_xor(cb[e], 32); }
  if (lane < 16 && ok) { SeqVec<CW>::st(sP + (wave * 2) * DP + c0, cg); SeqVec<CW>::st(sP + (wave * 2 + 1) * DP + c0, cb); }
  if (ok) {
    float v[CW], vm[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) {
      v[e] = rs * (g[e] * ga[e] - m1 - xh[e] * m2);
      vm[e] = v[e] * seq_dmul(dk, key, idx0 + o + e);
    }
    SeqVec<CW>::st(sDz + row * str + c0, v);
    SeqVec<CW>::st(sDzm + row * str + c0, vm);
    if (gdz) SeqVec<CW>::st(gdz + o, v);
    if (gdzm) SeqVec<CW>::st(gdzm + o, vm);
  }
}
// ... after the barrier: dgamma / dbeta partials of this sequence -> part[2][d], waves summed in a fixed order.  Runs on the LAST
// 2 * DP threads of the workgroup (the waves with the least matmul work in the stage that follows).
template <int DP>
__device__ __forceinline__ void seq_ln_part(const float* sP, float* part, const int d, const int tid) {
  const int t = tid - (GT_SEQ_NT - 2 * DP);
  if (t < 0) return;
  const int which = t / DP, c = t % DP;
  if (c >= d) return;
  float s = sP[which * DP + c];
#pragma unroll
  for (int w = 1; w < GT_SEQ_WAVES; ++w) s += sP[(2 * w + which) * DP + c];
  part[which * d + c] = s;
}

// ================================================================================================================ attention
// The transposed-score scheme of gt_attn.h (S^T = K Q^T so that softmax rows are in-lane and P feeds the next MFMA without any
// data movement) on LDS operands: q / k / v (and dctx in the backward) are tiles of this workgroup, read with 32-bit LDS
// addresses.  Wave pair p = wave >> 1 takes head h4 + p, wave & 1 the query (role 2: key) tile.
// HDC: head-dim class the kernel is compiled for -- 0: head_dim < 16 (operands zero-padded to 16 columns), else 16 / 32 / 64
template <int HDC> struct SeqHd { static constexpr int HD = HDC ? HDC : 16; static constexpr bool PAD = HDC == 0; };
// PAD: columns >= head_dim read as 0 (the address is clamped to the head's first column: always inside the tile)
template <bool PAD>
__device__ __forceinline__ float4 seq_ld4(const float* p, const int col, const int hd) {
  if (!PAD) return *reinterpret_cast<const float4*>(p);
  float4 v;
  v.x = col + 0 < hd ? p[0] : 0.f; v.y = col + 1 < hd ? p[col + 1 < hd ? 1 : 0] : 0.f;
  v.z = col + 2 < hd ? p[col + 2 < hd ? 2 : 0] : 0.f; v.w = col + 3 < hd ? p[col + 3 < hd ? 3 : 0] : 0.f;
  if (!(col < hd)) v.x = 0.f;
  return v;
}
template <bool PAD>
__device__ __forceinline__ float seq_ld1(const float* p, const int col, const int hd) {
  if (!PAD) return *p;
  const float v = p[col < hd ? 0 : -col];
  return col < hd ? v : 0.f;
}
struct SeqAttn {
  const float* q; const float* k; const float* v; int ldq;      // LDS tiles (row stride ldq), already offset to the head's first column
  float* P; uint32_t pidx;                                       // this head's probabilities (global, [32][32]) and its dropout index base
  int hd; float scale;
};
template <int HD, bool PAD>
__device__ __forceinline__ void seq_attn_fwd(const SeqAttn& a, float* ctx, const int ldc, const SeqDropK& dk, const uint32_t key, const int ti,
                                             const int lane) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  const int i = 16 * ti + l16;                                   // this lane's query row
  const float* qrow = a.q + i * a.ldq + (PAD ? 0 : 4 * g);
  const float* krow = a.k + l16 * a.ldq + (PAD ? 0 : 4 * g);     // key tile 0; tile 1 = + 16 rows
  float4 qf[NQ], k0[NQ], k1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (!PAD) {
      qf[q] = *reinterpret_cast<const float4*>(qrow + 16 * q);
      k0[q] = *reinterpret_cast<const float4*>(krow + 16 * q);
      k1[q] = *reinterpret_cast<const float4*>(krow + 16 * a.ldq + 16 * q);
    } else {                                                      // head_dim < 16: element 4 g + j of the head, 0 beyond it
      float t[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j, cc = c < hdr ? c : 0;
        t[0][j] = c < hdr ? qrow[cc] : 0.f; t[1][j] = c < hdr ? krow[cc] : 0.f; t[2][j] = c < hdr ? krow[16 * a.ldq + cc] : 0.f;
      }
      qf[q] = make_float4(t[0][0], t[0][1], t[0][2], t[0][3]);
      k0[q] = make_float4(t[1][0], t[1][1], t[1][2], t[1][3]);
      k1[q] = make_float4(t[2][0], t[2][1], t[2][2], t[2][3]);
    }
  }
  float vb[NQ][2][4];                                             // V[4g + c + 16 tj][16 ct + l16]
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 16 * ct + l16, cc = (!PAD || col < hdr) ? col : 0;
        const float v = a.v[(4 * g + c + 16 * tj) * a.ldq + cc];
        vb[ct][tj][c] = (!PAD || col < hdr) ? v : 0.f;
      }
  f32x4 st[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};           // S^T tiles [tj]
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    st[0] = GT_MFMA16(k0[q].x, qf[q].x, st[0]); st[1] = GT_MFMA16(k1[q].x, qf[q].x, st[1]);
    st[0] = GT_MFMA16(k0[q].y, qf[q].y, st[0]); st[1] = GT_MFMA16(k1[q].y, qf[q].y, st[1]);
    st[0] = GT_MFMA16(k0[q].z, qf[q].z, st[0]); st[1] = GT_MFMA16(k1[q].z, qf[q].z, st[1]);
    st[0] = GT_MFMA16(k0[q].w, qf[q].w, st[0]); st[1] = GT_MFMA16(k1[q].w, qf[q].w, st[1]);
  }
  float sv[2][4], mx = -INFINITY;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) { sv[tj][r] = st[tj][r] * a.scale; mx = fmaxf(mx, sv[tj][r]); }
  mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) { sv[tj][r] = expf(sv[tj][r] - mx); sum += sv[tj][r]; }
  sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  float pd[2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const unsigned o = (unsigned)(i * 32 + 16 * tj + 4 * g);
    float4 pv;
    pv.x = sv[tj][0] * inv; pv.y = sv[tj][1] * inv; pv.z = sv[tj][2] * inv; pv.w = sv[tj][3] * inv;
    *reinterpret_cast<float4*>(a.P + o) = pv;
    pd[tj][0] = pv.x * seq_dmul(dk, key, a.pidx + o);
    pd[tj][1] = pv.y * seq_dmul(dk, key, a.pidx + o + 1);
    pd[tj][2] = pv.z * seq_dmul(dk, key, a.pidx + o + 2);
    pd[tj][3] = pv.w * seq_dmul(dk, key, a.pidx + o + 3);
  }
  f32x4 o[NQ];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct) {
    o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) o[ct] = GT_MFMA16(pd[tj][c], vb[ct][tj][c], o[ct]);
  }
  float* orow = ctx + (16 * ti + 4 * g) * ldc + l16;
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { if (!PAD || 16 * ct + l16 < hdr) orow[r * ldc + 16 * ct] = o[ct][r]; }
}

// Backward, two roles per wave with a workgroup barrier between them (gt_attn.h): role 1 (query tile w) -> dq in registers and
// the row sums rd -> srd (32 floats of LDS per head); role 2 (key tile w) -> dk, dv, and the dq / dk / dv stores into the dqkv tile.
template <int HD, bool PAD>
__device__ __forceinline__ void seq_attn_bwd1(const SeqAttn& a, const float* dctx, const int lddc, const SeqDropK& dk, const uint32_t key,
                                              const int w, const int lane, float* srd, f32x4 (&dq_out)[HD / 16]) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  const int i = 16 * w + l16;
  const float* dorow = dctx + i * lddc + 4 * g;
  const float* vrow = a.v + l16 * a.ldq + 4 * g;
  float4 df[NQ], v0[NQ], v1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    df[q] = seq_ld4<PAD>(dorow + 16 * q, 16 * q + 4 * g, hdr);
    v0[q] = seq_ld4<PAD>(vrow + 16 * q, 16 * q + 4 * g, hdr);
    v1[q] = seq_ld4<PAD>(vrow + 16 * a.ldq + 16 * q, 16 * q + 4 * g, hdr);
  }
  const float* kcol = a.k + 4 * g * a.ldq + l16;
  float kb[NQ][2][4];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) kb[ct][tj][c] = seq_ld1<PAD>(kcol + (16 * tj + c) * a.ldq + 16 * ct, 16 * ct + l16, hdr);
  float4 pv[2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) pv[tj] = *reinterpret_cast<const float4*>(a.P + (unsigned)(i * 32 + 16 * tj + 4 * g));
  f32x4 dt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};           // dPd^T tiles [tj]
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    dt[0] = GT_MFMA16(v0[q].x, df[q].x, dt[0]); dt[1] = GT_MFMA16(v1[q].x, df[q].x, dt[1]);
    dt[0] = GT_MFMA16(v0[q].y, df[q].y, dt[0]); dt[1] = GT_MFMA16(v1[q].y, df[q].y, dt[1]);
    dt[0] = GT_MFMA16(v0[q].z, df[q].z, dt[0]); dt[1] = GT_MFMA16(v1[q].z, df[q].z, dt[1]);
    dt[0] = GT_MFMA16(v0[q].w, df[q].w, dt[0]); dt[1] = GT_MFMA16(v1[q].w, df[q].w, dt[1]);
  }
  float p[2][4], dp[2][4], rd = 0.f;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const uint32_t idx0 = a.pidx + (uint32_t)(i * 32 + 16 * tj + 4 * g);
    p[tj][0] = pv[tj].x; p[tj][1] = pv[tj].y; p[tj][2] = pv[tj].z; p[tj][3] = pv[tj].w;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dp[tj][r] = dt[tj][r] * seq_dmul(dk, key, idx0 + r);
      rd += dp[tj][r] * p[tj][r];
    }
  }
  rd += __shfl_xor(rd, 16); rd += __shfl_xor(rd, 32);
  if (g == 0) srd[i] = rd;
  float ds[2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[tj][r] = p[tj][r] * (dp[tj][r] - rd) * a.scale;
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct) {
    dq_out[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) dq_out[ct] = GT_MFMA16(ds[tj][c], kb[ct][tj][c], dq_out[ct]);
  }
}
template <int HD, bool PAD>
__device__ __forceinline__ void seq_attn_bwd2(const SeqAttn& a, const float* dctx, const int lddc, const SeqDropK& dk, const uint32_t key,
                                              const int w, const int lane, const float* srd, const f32x4 (&dq_out)[HD / 16], float* dq,
                                              const int lddq, const int dstep) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  const int j = 16 * w + l16;
  const float* dorow = dctx + l16 * lddc + 4 * g;                 // query tile 0; tile 1 = + 16 rows
  const float* vrow = a.v + j * a.ldq + 4 * g;
  float4 vf[NQ], d0[NQ], d1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    vf[q] = seq_ld4<PAD>(vrow + 16 * q, 16 * q + 4 * g, hdr);
    d0[q] = seq_ld4<PAD>(dorow + 16 * q, 16 * q + 4 * g, hdr);
    d1[q] = seq_ld4<PAD>(dorow + 16 * lddc + 16 * q, 16 * q + 4 * g, hdr);
  }
  const float* docol = dctx + 4 * g * lddc + l16;
  const float* qcol = a.q + 4 * g * a.ldq + l16;
  float db[NQ][2][4], qb[NQ][2][4];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        db[ct][ti][c] = seq_ld1<PAD>(docol + (16 * ti + c) * lddc + 16 * ct, 16 * ct + l16, hdr);
        qb[ct][ti][c] = seq_ld1<PAD>(qcol + (16 * ti + c) * a.ldq + 16 * ct, 16 * ct + l16, hdr);
      }
  float pvv[2][4], rdv[2][4];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ti + 4 * g + r;
      pvv[ti][r] = a.P[(unsigned)(i * 32 + j)];
      rdv[ti][r] = srd[i];
    }
  f32x4 dd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};           // dPd tiles [ti]
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    dd[0] = GT_MFMA16(d0[q].x, vf[q].x, dd[0]); dd[1] = GT_MFMA16(d1[q].x, vf[q].x, dd[1]);
    dd[0] = GT_MFMA16(d0[q].y, vf[q].y, dd[0]); dd[1] = GT_MFMA16(d1[q].y, vf[q].y, dd[1]);
    dd[0] = GT_MFMA16(d0[q].z, vf[q].z, dd[0]); dd[1] = GT_MFMA16(d1[q].z, vf[q].z, dd[1]);
    dd[0] = GT_MFMA16(d0[q].w, vf[q].w, dd[0]); dd[1] = GT_MFMA16(d1[q].w, vf[q].w, dd[1]);
  }
  float pdm[2][4], ds[2][4];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ti + 4 * g + r;
      const float mk = seq_dmul(dk, key, a.pidx + (uint32_t)(i * 32 + j));
      pdm[ti][r] = pvv[ti][r] * mk;
      ds[ti][r] = pvv[ti][r] * (dd[ti][r] * mk - rdv[ti][r]) * a.scale;
    }
  f32x4 ov[NQ], ok[NQ];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct) {
    ov[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; ok[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ov[ct] = GT_MFMA16(pdm[ti][c], db[ct][ti][c], ov[ct]);
        ok[ct] = GT_MFMA16(ds[ti][c], qb[ct][ti][c], ok[ct]);
      }
  }
  float* dqrow = dq + (16 * w + 4 * g) * lddq + l16;              // dq; dk = + dstep columns, dv = + 2 dstep
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (PAD && 16 * ct + l16 >= hdr) continue;
      dqrow[r * lddq + 16 * ct] = dq_out[ct][r];
      dqrow[r * lddq + 16 * ct + dstep] = ok[ct][r];
      dqrow[r * lddq + 16 * ct + 2 * dstep] = ov[ct][r];
    }
}

// ================================================================================================================ forward
template <int DP, int HDC>
__global__ __launch_bounds__(GT_SEQ_NT) void seq_fwd_kernel(SeqArgs a) {
  constexpr int SX = DP + 8, SH = GT_SEQ_FMAX + 8, SQ = 3 * DP + 8, SRS = DP + 8, RP = (DP == 32) ? 8 : 2, CW = DP / 16;
  constexpr int HD = SeqHd<HDC>::HD;
  constexpr bool PAD = SeqHd<HDC>::PAD;
  __shared__ __attribute__((aligned(16))) float sX[32 * SX], sX1[32 * SX], sC[32 * SX], sQ[32 * SQ], sH[32 * SH], sR[RP * 32 * SRS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x, d = a.d, F = a.F;
  const size_t r0 = (size_t)b * 32;                          // first token row of this sequence
  const float* const zp = gt_zero_ptr();
  const float* prm = a.prm;
  float* ws = a.ws;
  const SeqDropK dk = seq_dropk(a);
  const uint32_t idxd = (uint32_t)(r0 * d), idxf = (uint32_t)(r0 * F);   // dropout element index of this sequence's first row

  GT_STAMP(0);
  // ---- input layer: a0 = x Win^T + b; x0 = drop(relu(a0) + pe)     (A tile: the 32 x S input rows, zero-padded to 32 columns)
  for (int e = tid; e < 32 * 32; e += GT_SEQ_NT) {
    const int r = e >> 5, c = e & 31;
    sC[r * SX + c] = *(c < a.S ? a.xin + (r0 + r) * a.S + c : zp);
  }
  GT_BARRIER();
  GT_STAMP(1);
  {
    const uint32_t key = seq_key(dk, GT_SITE_PE_ENC);
    float* ga0 = ws + a.a0 + r0 * d;
    float* gx0 = ws + a.x0 + r0 * d;
    seq_mm_edge<false, false>(sC, SX, a.S, prm + a.in_w, a.S, d, prm + a.in_b, wave, lane, zp,
                              [&](int n0, const f32x4& c0, const f32x4& c1, const float (&bi)[4]) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int row = 16 * h2 + l16, col = n0 + 4 * lg;
        const f32x4& c = h2 ? c1 : c0;
        const float4 pe = *reinterpret_cast<const float4*>(a.pe + row * d + col);
        const unsigned o = (unsigned)(row * d + col);
        const float4 pre = make_float4(c[0] + bi[0], c[1] + bi[1], c[2] + bi[2], c[3] + bi[3]);
        float4 v;
        v.x = (fmaxf(pre.x, 0.f) + pe.x) * seq_dmul(dk, key, idxd + o);
        v.y = (fmaxf(pre.y, 0.f) + pe.y) * seq_dmul(dk, key, idxd + o + 1);
        v.z = (fmaxf(pre.z, 0.f) + pe.z) * seq_dmul(dk, key, idxd + o + 2);
        v.w = (fmaxf(pre.w, 0.f) + pe.w) * seq_dmul(dk, key, idxd + o + 3);
        *reinterpret_cast<float4*>(ga0 + o) = pre;
        *reinterpret_cast<float4*>(gx0 + o) = v;
        *reinterpret_cast<float4*>(&sX[row * SX + col]) = v;
      }
    });
  }
  GT_BARRIER();

  const float ascale = 1.0f / sqrtf((float)a.hd);
  for (int l = 0; l < a.L; ++l) {
    const float* pl = prm + (int64_t)l * a.pstride;          // this layer's parameters / saved activations (wave-uniform bases)
    float* wl = ws + (int64_t)l * a.wstride;
    const int site0 = GT_SITE_LAYER0 + 8 * l;
    const int sb = 2 + 10 * l;                               // stamp base of this layer
    GT_STAMP(sb);
    // ---- in-proj: qkv = x Win^T + b -> LDS (the attention bodies read it there) and global (saved for backward)
    {
      float* gq = wl + a.w0.qkv + r0 * 3 * d;
      seq_mm_tiles<false, (3 * DP / 16 + 7) / 8>(sX, SX, d, pl + a.p0.in_w, d, 3 * d, pl + a.p0.in_b, wave, lane,
                                                 [&](int n0, const f32x4& c0, const f32x4& c1, const float4& bi) {
        const int col = n0 + 4 * lg;
        const float4 o0 = make_float4(c0[0] + bi.x, c0[1] + bi.y, c0[2] + bi.z, c0[3] + bi.w);
        const float4 o1 = make_float4(c1[0] + bi.x, c1[1] + bi.y, c1[2] + bi.z, c1[3] + bi.w);
        *reinterpret_cast<float4*>(&sQ[l16 * SQ + col]) = o0;
        *reinterpret_cast<float4*>(&sQ[(16 + l16) * SQ + col]) = o1;
        *reinterpret_cast<float4*>(gq + (unsigned)(l16 * 3 * d + col)) = o0;
        *reinterpret_cast<float4*>(gq + (unsigned)((16 + l16) * 3 * d + col)) = o1;
      });
    }
    GT_BARRIER();
    GT_STAMP(sb + 1);
    // ---- attention: four heads at a time (wave pair p = wave >> 1 takes head h4 + p); operands from the LDS qkv tile,
    // P to global, ctx to the LDS tile
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_ATTN);
      for (int h4 = 0; h4 < a.H; h4 += GT_SEQ_WAVES / 2) {
        const int h = h4 + (wave >> 1);
        if (h < a.H) {
          SeqAttn at;
          at.q = sQ + h * a.hd; at.k = at.q + d; at.v = at.q + 2 * d; at.ldq = SQ; at.hd = a.hd; at.scale = ascale;
          at.pidx = (uint32_t)((b * a.H + h) * 1024); at.P = wl + a.w0.P + (size_t)(b * a.H + h) * 1024;
          seq_attn_fwd<HD, PAD>(at, sC + h * a.hd, SX, dk, key, wave & 1, lane);
        }
      }
    }
    GT_BARRIER();
    GT_STAMP(sb + 2);
    // ---- out-proj (raw product -> sR part 0); the ctx tile also goes to global here (operand of the out-proj weight gradient)
    {
      float* gc = wl + a.w0.ctx + r0 * d;
      const int row = tid >> 4, c0 = (tid & 15) * CW;
      if (c0 < d) { float v[CW]; SeqVec<CW>::ld(v, &sC[row * SX + c0]); SeqVec<CW>::st(gc + (unsigned)(row * d + c0), v); }
      seq_mm_square<false>(sC, SX, d, pl + a.p0.out_w, sR, SRS, wave, lane);
    }
    GT_BARRIER();
    GT_STAMP(sb + 3);
    // ---- z1 = drop(. + b_o) + x;  norm1 -> x1
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROP1);
      const float* bo = pl + a.p0.out_b;
      seq_ln_fwd<DP>([&](int row, int c0, float (&z)[CW]) {
        float bi[CW], xr[CW];
        SeqVec<CW>::ld(z, &sR[row * SRS + c0]); SeqVec<CW>::ld(bi, bo + c0); SeqVec<CW>::ld(xr, &sX[row * SX + c0]);
#pragma unroll
        for (int e = 0; e < CW; ++e) z[e] = (z[e] + bi[e]) * seq_dmul(dk, key, idxd + (uint32_t)(row * d + c0 + e)) + xr[e];
      }, sX1, SX, d, pl + a.p0.n1w, pl + a.p0.n1b, wl + a.w0.x1 + r0 * d, wl + a.w0.xhat1 + r0 * d, wl + a.w0.rstd1 + r0, tid);
    }
    GT_BARRIER();
    GT_STAMP(sb + 4);
    // ---- FFN1: hact = drop(relu(x1 W1^T + b1))
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_FFN);
      float* gh = wl + a.w0.hact + r0 * F;
      seq_mm_tiles<false, GT_SEQ_FMAX / 128>(sX1, SX, d, pl + a.p0.w1, d, F, pl + a.p0.b1, wave, lane,
                                             [&](int n0, const f32x4& c0, const f32x4& c1, const float4& bi) {
        const int col = n0 + 4 * lg;
        const unsigned o0 = (unsigned)(l16 * F + col), o1 = (unsigned)((16 + l16) * F + col);
        float4 v0, v1;
        v0.x = fmaxf(c0[0] + bi.x, 0.f) * seq_dmul(dk, key, idxf + o0);     v1.x = fmaxf(c1[0] + bi.x, 0.f) * seq_dmul(dk, key, idxf + o1);
        v0.y = fmaxf(c0[1] + bi.y, 0.f) * seq_dmul(dk, key, idxf + o0 + 1); v1.y = fmaxf(c1[1] + bi.y, 0.f) * seq_dmul(dk, key, idxf + o1 + 1);
        v0.z = fmaxf(c0[2] + bi.z, 0.f) * seq_dmul(dk, key, idxf + o0 + 2); v1.z = fmaxf(c1[2] + bi.z, 0.f) * seq_dmul(dk, key, idxf + o1 + 2);
        v0.w = fmaxf(c0[3] + bi.w, 0.f) * seq_dmul(dk, key, idxf + o0 + 3); v1.w = fmaxf(c1[3] + bi.w, 0.f) * seq_dmul(dk, key, idxf + o1 + 3);
        *reinterpret_cast<float4*>(&sH[l16 * SH + col]) = v0;
        *reinterpret_cast<float4*>(&sH[(16 + l16) * SH + col]) = v1;
        *reinterpret_cast<float4*>(gh + o0) = v0;
        *reinterpret_cast<float4*>(gh + o1) = v1;
      });
    }
    GT_BARRIER();
    GT_STAMP(sb + 5);
    // ---- FFN2 (K = F: split over the waves) -> partial tiles
    seq_mm_splitk<false>(sH, SH, F, pl + a.p0.w2, F, d, sR, SRS, wave, lane);
    GT_BARRIER();
    GT_STAMP(sb + 6);
    // ---- z2 = drop(sum of the parts + b2) + x1;  norm2 -> the next layer's input
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROPF);
      const float* b2 = pl + a.p0.b2;
      const int parts = seq_splitk_parts(d);
      seq_ln_fwd<DP>([&](int row, int c0, float (&z)[CW]) {
        float bi[CW], xr[CW];
        seq_parts_sum<CW>(z, sR, SRS, parts, row, c0); SeqVec<CW>::ld(bi, b2 + c0); SeqVec<CW>::ld(xr, &sX1[row * SX + c0]);
#pragma unroll
        for (int e = 0; e < CW; ++e) z[e] = (z[e] + bi[e]) * seq_dmul(dk, key, idxd + (uint32_t)(row * d + c0 + e)) + xr[e];
      }, sX, SX, d, pl + a.p0.n2w, pl + a.p0.n2b, wl + a.w0.xout + r0 * d, wl + a.w0.xhat2 + r0 * d, wl + a.w0.rstd2 + r0, tid);
    }
    GT_BARRIER();
    GT_STAMP(sb + 7);
  }
  // ---- final encoder norm -> memory, then the output layer: [h logits | sigmoid v | 0.5 tanh o]
  seq_ln_fwd<DP>([&](int row, int c0, float (&z)[CW]) { SeqVec<CW>::ld(z, &sX[row * SX + c0]); }, sC, SX, d, prm + a.encn_w, prm + a.encn_b,
                 ws + a.memory + r0 * d, ws + a.enc_xhat + r0 * d, ws + a.enc_rstd + r0, tid);
  GT_BARRIER();
  seq_mm_edge<false, true>(sC, SX, d, prm + a.out_w, d, GT_TGT, prm + a.out_b, wave, lane, zp,
                           [&](int n0, const f32x4& c0, const f32x4& c1, const float (&bi)[4]) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const f32x4& c = h2 ? c1 : c0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = n0 + 4 * lg + r;
        if (col < GT_TGT) {
          float v = c[r] + bi[r];
          if (col >= 2 * GT_VOICES) v = 0.5f * tanhf(v);
          else if (col >= GT_VOICES) v = gt_sigmoid(v);
          a.hvo[(r0 + 16 * h2 + l16) * GT_TGT + col] = v;
        }
      }
    }
  });
  GT_STAMP(2 + 10 * a.L);
}

// ================================================================================================================ backward
// LayerNorm jobs (dgamma / dbeta partial blocks, [B][2][d] each) in the order the kernel fills them: 0 = final norm, then for
// l = L-1 .. 0: 1 + 2 (L-1-l) = norm2 of layer l, 2 + 2 (L-1-l) = norm1 of layer l.  The host registers them in this order.
template <int DP, int HDC>
__global__ __launch_bounds__(GT_SEQ_NT) void seq_bwd_kernel(SeqArgs a) {
  constexpr int SX = DP + 8, SH = GT_SEQ_FMAX + 8, SQ = 3 * DP + 8, SRS = DP + 8, RP = (DP == 32) ? 8 : 2, CW = DP / 16;
  constexpr int HD = SeqHd<HDC>::HD;
  constexpr bool PAD = SeqHd<HDC>::PAD;
  // sG: gradient w.r.t. the current layer's output; sDZ: the LayerNorm backward's dz (residual gradient); sC: dz * dropout mask
  // (A operand of the next dgrad); sZ: dctx; sQ: the dqkv tile; sH: the FFN tile -- and, between the FFN1 dgrad and the end of
  // the attention backward, the saved qkv tile of this layer (sK = sH, [32][SQ])
  __shared__ __attribute__((aligned(16))) float sG[32 * SX], sZ[32 * SX], sDZ[32 * SX], sC[32 * SX], sQ[32 * SQ], sH[32 * SH], sR[RP * 32 * SRS],
      sP[2 * GT_SEQ_WAVES * DP], srd[32 * GT_SEQ_WAVES / 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x, d = a.d, F = a.F;
  const size_t r0 = (size_t)b * 32;
  const float* const zp = gt_zero_ptr();
  const float* prm = a.prm;
  float* ws = a.ws;
  const SeqDropK dk = seq_dropk(a);
  const uint32_t idxd = (uint32_t)(r0 * d);
  const float mscale = dk.thr ? dk.scale : 1.0f;
  const float ascale = 1.0f / sqrtf((float)a.hd);
  float* const sK = sH;
  auto part_at = [&](int job) { return ws + a.ln_part + (int64_t)job * a.ln_part_stride + (size_t)b * 2 * d; };
  // a [32][ncol] tile of this sequence, global -> LDS, 16 bytes per thread and pass
  auto load_tile = [&](float* dst, const int str, const float* src, const int ncol) {
    const int q4 = ncol >> 2;
    for (int e = tid; e < 32 * q4; e += GT_SEQ_NT) {
      const int r = e / q4, c = (e - r * q4) * 4;
      *reinterpret_cast<float4*>(dst + r * str + c) = *reinterpret_cast<const float4*>(src + (unsigned)(r * ncol + c));
    }
  };

  GT_STAMP(100);
  // ---- output layer dgrad: dmem = dlogits Wout   (A tile: 32 x 27 zero-padded to 32 columns)
  for (int e = tid; e < 32 * 32; e += GT_SEQ_NT) {
    const int r = e >> 5, c = e & 31;
    sC[r * SX + c] = *(c < GT_TGT ? ws + a.dlogits + (r0 + r) * GT_TGT + c : zp);
  }
  load_tile(sH, SH, ws + a.w0.hact + (int64_t)(a.L - 1) * a.wstride + r0 * F, F);
  GT_BARRIER();
  seq_mm_edge<true, true>(sC, SX, GT_TGT, prm + a.out_w, d, d, nullptr, wave, lane, zp,
                          [&](int n0, const f32x4& c0, const f32x4& c1, const float (&)[4]) {
    const int col = n0 + 4 * lg;
    *reinterpret_cast<float4*>(&sZ[l16 * SX + col]) = make_float4(c0[0], c0[1], c0[2], c0[3]);
    *reinterpret_cast<float4*>(&sZ[(16 + l16) * SX + col]) = make_float4(c1[0], c1[1], c1[2], c1[3]);
  });
  GT_BARRIER();
  // ---- final norm backward -> gradient w.r.t. the last layer's output (sG); its dz is not a weight-gradient operand
  {
    SeqDropK nd = dk; nd.thr = 0u;
    seq_ln_bwd<DP>([&](int row, int c0, float (&g)[CW]) { SeqVec<CW>::ld(g, &sZ[row * SX + c0]); }, sG, sC, SX, d, ws + a.enc_xhat + r0 * d,
                   ws + a.enc_rstd + r0, prm + a.encn_w, nd, 0u, 0u, nullptr, nullptr, sP, tid);
  }
  GT_BARRIER();
  seq_ln_part<DP>(sP, part_at(0), d, tid);
  GT_BARRIER();                                                 // (sP is rewritten by the first norm2 backward below)
  GT_STAMP(101);

  bool first = true;                                            // the first norm2 backward reads g from sG; later ones add the parts
  for (int l = a.L - 1; l >= 0; --l) {
    const float* pl = prm + (int64_t)l * a.pstride;
    float* wl = ws + (int64_t)l * a.wstride;
    float* tl = ws + (int64_t)l * a.tstride;
    const int site0 = GT_SITE_LAYER0 + 8 * l, jb = 1 + 2 * (a.L - 1 - l);
    const int sb = 102 + 10 * (a.L - 1 - l);
    // ---- norm2 backward: g = gradient w.r.t. this layer's output (l == L-1: sG; else the in-proj dgrad parts of layer l+1 + its
    // dz1) -> dz2 -> sDZ, dz2 * mask(dropout on the FFN output) -> sC; both to global for the weight gradients
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROPF);
      const int parts = seq_splitk_parts(d);
      const bool fromg = first;
      seq_ln_bwd<DP>([&](int row, int c0, float (&g)[CW]) {
        if (fromg) SeqVec<CW>::ld(g, &sG[row * SX + c0]);
        else {
          float r[CW];
          seq_parts_sum<CW>(g, sR, SRS, parts, row, c0); SeqVec<CW>::ld(r, &sDZ[row * SX + c0]);
#pragma unroll
          for (int e = 0; e < CW; ++e) g[e] += r[e];
        }
      }, sDZ, sC, SX, d, wl + a.w0.xhat2 + r0 * d, wl + a.w0.rstd2 + r0, pl + a.p0.n2w, dk, key, idxd, tl + a.t0.dzA + r0 * d,
                     dk.thr ? tl + a.t0.dzAm + r0 * d : nullptr, sP, tid);
      first = false;
    }
    GT_BARRIER();
    GT_STAMP(sb);
    // ---- FFN2 dgrad: dhid = (dz2m W2) * [hact != 0] * 1/(1-p), in place over the hact tile in sH
    seq_ln_part<DP>(sP, part_at(jb), d, tid);
    {
      float* gd = tl + a.t0.dhid + r0 * F;
      seq_mm_tiles<true, GT_SEQ_FMAX / 128>(sC, SX, d, pl + a.p0.w2, F, F, nullptr, wave, lane,
                                            [&](int n0, const f32x4& c0, const f32x4& c1, const float4&) {
        const int col = n0 + 4 * lg;
        const float4 ha = *reinterpret_cast<const float4*>(&sH[l16 * SH + col]), hb = *reinterpret_cast<const float4*>(&sH[(16 + l16) * SH + col]);
        const float4 o0 = make_float4(ha.x != 0.f ? c0[0] * mscale : 0.f, ha.y != 0.f ? c0[1] * mscale : 0.f, ha.z != 0.f ? c0[2] * mscale : 0.f,
                                      ha.w != 0.f ? c0[3] * mscale : 0.f);
        const float4 o1 = make_float4(hb.x != 0.f ? c1[0] * mscale : 0.f, hb.y != 0.f ? c1[1] * mscale : 0.f, hb.z != 0.f ? c1[2] * mscale : 0.f,
                                      hb.w != 0.f ? c1[3] * mscale : 0.f);
        *reinterpret_cast<float4*>(&sH[l16 * SH + col]) = o0;
        *reinterpret_cast<float4*>(&sH[(16 + l16) * SH + col]) = o1;
        *reinterpret_cast<float4*>(gd + (unsigned)(l16 * F + col)) = o0;
        *reinterpret_cast<float4*>(gd + (unsigned)((16 + l16) * F + col)) = o1;
      });
    }
    GT_BARRIER();
    GT_STAMP(sb + 1);
    // ---- FFN1 dgrad (K = F: split over the waves) -> partial tiles
    seq_mm_splitk<true>(sH, SH, F, pl + a.p0.w1, d, d, sR, SRS, wave, lane);
    GT_BARRIER();
    GT_STAMP(sb + 2);
    // ---- norm1 backward: g1 = parts + dz2 -> dz1 -> sDZ, dz1 * mask(dropout1) -> sC.  sH is free from here to the end of the
    // attention backward: the saved qkv tile of this layer is requested into it now
    load_tile(sK, SQ, wl + a.w0.qkv + r0 * 3 * d, 3 * d);
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROP1);
      const int parts = seq_splitk_parts(d);
      seq_ln_bwd<DP>([&](int row, int c0, float (&g)[CW]) {
        float r[CW];
        seq_parts_sum<CW>(g, sR, SRS, parts, row, c0); SeqVec<CW>::ld(r, &sDZ[row * SX + c0]);
#pragma unroll
        for (int e = 0; e < CW; ++e) g[e] += r[e];
      }, sDZ, sC, SX, d, wl + a.w0.xhat1 + r0 * d, wl + a.w0.rstd1 + r0, pl + a.p0.n1w, dk, key, idxd, tl + a.t0.dzB + r0 * d,
                     dk.thr ? tl + a.t0.dzBm + r0 * d : nullptr, sP, tid);
    }
    GT_BARRIER();
    GT_STAMP(sb + 3);
    // ---- out-proj dgrad: dctx = dz1m Wo -> sZ (LDS: the attention backward reads it there)
    seq_ln_part<DP>(sP, part_at(jb + 1), d, tid);
    seq_mm_square<true>(sC, SX, d, pl + a.p0.out_w, sZ, SX, wave, lane);
    GT_BARRIER();
    GT_STAMP(sb + 4);
    // ---- attention backward, four heads at a time: q / k / v from the LDS copy, P from global (saved), dctx from LDS,
    // dq / dk / dv -> the sQ tile
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_ATTN);
      for (int h4 = 0; h4 < a.H; h4 += GT_SEQ_WAVES / 2) {
        const int h = h4 + (wave >> 1);
        const bool active = h < a.H;
        SeqAttn at;
        at.q = sK + h * a.hd; at.k = at.q + d; at.v = at.q + 2 * d; at.ldq = SQ; at.hd = a.hd; at.scale = ascale;
        at.pidx = (uint32_t)((b * a.H + h) * 1024); at.P = wl + a.w0.P + (size_t)(b * a.H + h) * 1024;
        f32x4 dq_out[HD / 16];
        if (active) seq_attn_bwd1<HD, PAD>(at, sZ + h * a.hd, SX, dk, key, wave & 1, lane, srd + 32 * (wave >> 1), dq_out);
        GT_BARRIER();
        if (active) seq_attn_bwd2<HD, PAD>(at, sZ + h * a.hd, SX, dk, key, wave & 1, lane, srd + 32 * (wave >> 1), dq_out, sQ + h * a.hd, SQ, d);
        GT_BARRIER();
      }
    }
    GT_STAMP(sb + 5);
    // ---- dqkv tile -> global (operand of the in-proj weight gradient); in-proj dgrad (K = 3 d: split over the waves); the
    // FFN tile of the next layer down is requested into sH (the qkv copy in it is dead now)
    {
      float* gq = tl + a.t0.dqkv + r0 * 3 * d;
      const int q4 = (3 * d) >> 2;
      for (int e = tid; e < 32 * q4; e += GT_SEQ_NT) {
        const int r = e / q4, c = (e - r * q4) * 4;
        *reinterpret_cast<float4*>(gq + (unsigned)(r * 3 * d + c)) = *reinterpret_cast<const float4*>(&sQ[r * SQ + c]);
      }
    }
    if (l > 0) load_tile(sH, SH, ws + (int64_t)(l - 1) * a.wstride + a.w0.hact + r0 * F, F);
    seq_mm_splitk<true>(sQ, SQ, 3 * d, pl + a.p0.in_w, d, d, sR, SRS, wave, lane);
    GT_BARRIER();
    GT_STAMP(sb + 6);
  }
  // ---- InputLayer backward: da0 = (parts + dz1) * dropout mask * [a0 > 0] -> global (operand of the input layer's weight gradient)
  {
    const uint32_t key = seq_key(dk, GT_SITE_PE_ENC);
    const int parts = seq_splitk_parts(d);
    const int row = tid >> 4, c0 = (tid & 15) * CW;
    if (c0 < d) {
      const unsigned o = (unsigned)(row * d + c0);
      float g[CW], r[CW], a0v[CW];
      seq_parts_sum<CW>(g, sR, SRS, parts, row, c0); SeqVec<CW>::ld(r, &sDZ[row * SX + c0]); SeqVec<CW>::ld(a0v, ws + a.a0 + r0 * d + o);
#pragma unroll
      for (int e = 0; e < CW; ++e) { const float v = (g[e] + r[e]) * seq_dmul(dk, key, idxd + o + e); g[e] = a0v[e] > 0.f ? v : 0.f; }
      SeqVec<CW>::st(ws + a.da0 + r0 * d + o, g);
    }
  }
  GT_STAMP(102 + 10 * a.L);
}
