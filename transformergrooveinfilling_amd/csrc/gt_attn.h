// T=32 multi-head attention core, forward and backward, one workgroup per (sequence, head).
//
// Follows torch:nn/functional.py:6504-6642: S = (q k^T)/sqrt(hd) [+ causal -inf], P = softmax(S),
// dropout(P), ctx = P v.  A whole 32x32 score tile lives in registers/LDS of one workgroup (8 lanes
// per query row), so there is no online softmax and nothing to shard.  The attention core is 2-5 %
// of the step's FLOPs; it is LDS-tiled VALU code, generic in head_dim (1..512) by
// walking head_dim in 32-column slabs.
// Saved for backward: P (pre-dropout probabilities, (B*H,32,32)); the dropout mask is regenerated
// from the counter-based hash.
//
// Two implementations with identical interfaces and results:
//   attn_*_mfma_kernel<HD>  head_dim 16/32/64/128: everything on v_mfma_f32_16x16x4_f32, operands straight from global
//                           memory into registers, no LDS tiles (see the layout notes above the kernels);
//   attn_*_kernel           any head_dim (1..512): LDS-tiled VALU code, 32-column slabs.
#pragma once
#include "gt_common.h"

struct AttnArgs {
  const float* q; const float* k; const float* v;   // row m, column h*hd + c (strides below)
  int ldq, ldk, ldv;
  float* P;                                          // (B*H, 32, 32)
  float* ctx; int ldc;                               // (M, d)
  int H, hd;
  float scale;
  int causal;
  DropArgs drop;
  // backward
  const float* dctx; int lddc;
  float* dq; float* dk; float* dv; int lddq, lddk, lddv;
  // bf16 copies for the GEMMs that take these outputs as operands at precision = 1 (GemmArgs::A16; the MFMA kernels only):
  // ctx16 (M, d) beside ctx; dqkv16 (M, 3 d) = [dq | dk | dv] beside a dqkv buffer the three gradients are written into.  nullptr: none
  uint16_t* ctx16; uint16_t* dqkv16;
  // precision = 2: q / k / v (and, backward, dctx) stored in bf16 ALONE -- same strides (in elements) as the fp32 pointers, which are then
  // unused.  Only the LDS-staged kernels (attn_fwd_lds_kernel / attn_bwd_lds_kernel) take them: the tiles are widened on their way into LDS
  // and the arithmetic is the fp32 one.
  const uint16_t* q16; const uint16_t* k16; const uint16_t* v16; const uint16_t* dctx16;
};

__device__ static inline void attn_load_slab(float (*s)[33], const float* src, int ld, int b, int h, int hd, int c0, int tid, const float* zp) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u, r = e >> 5, c = e & 31;
    const float* p = (c0 + c < hd) ? src + ((size_t)(b * 32 + r) * ld + h * hd + c0 + c) : zp;   // branch-free
    s[r][c] = *p;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  __shared__ float sq[32][33], sk[32][33], sp[32][33];
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int i = tid >> 3, jg = tid & 7;
  const float* const zp = gt_zero_ptr();
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < a.hd; c0 += 32) {
    attn_load_slab(sq, a.q, a.ldq, b, h, a.hd, c0, tid, zp);
    attn_load_slab(sk, a.k, a.ldk, b, h, a.hd, c0, tid, zp);
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
      const float qv = sq[i][c];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) s[jj] += qv * sk[jg + 8 * jj][c];
    }
    __syncthreads();
  }
  float mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jg + 8 * jj;
    s[jj] = (a.causal && j > i) ? -INFINITY : s[jj] * a.scale;
    mx = fmaxf(mx, s[jj]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
  float sum = 0.f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) { s[jj] = expf(s[jj] - mx); sum += s[jj]; }
  sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
  const float inv = 1.0f / sum;
  const uint32_t dkey = gt_drop_key(a.drop);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jg + 8 * jj;
    const float p = s[jj] * inv;
    const uint32_t idx = (uint32_t)((bh * 32 + i) * 32 + j);
    a.P[idx] = p;
    sp[i][j] = p * gt_drop_mul(a.drop, dkey, idx);
  }
  __syncthreads();
  for (int c0 = 0; c0 < a.hd; c0 += 32) {
    attn_load_slab(sk, a.v, a.ldv, b, h, a.hd, c0, tid, zp);
    __syncthreads();
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const float pv = sp[i][j];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) o[cc] += pv * sk[j][jg + 8 * cc];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = c0 + jg + 8 * cc;
      if (c < a.hd) a.ctx[(size_t)(b * 32 + i) * a.ldc + h * a.hd + c] = o[cc];
    }
    __syncthreads();
  }
}

// dPd = dctx v^T; dP = dPd*mask; dS = P*(dP - rowsum(dP*P))*scale; dv = (P*mask)^T dctx;
// dq = dS k; dk = dS^T q.   (masked / causal entries have P = 0, hence dS = 0.)
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
  __shared__ float sa[32][33], sb[32][33], sds[32][33], spd[32][33];
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int i = tid >> 3, jg = tid & 7;
  const float* const zp = gt_zero_ptr();
  const uint32_t dkey = gt_drop_key(a.drop);
  float p[4], mk[4], dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jg + 8 * jj;
    const uint32_t idx = (uint32_t)((bh * 32 + i) * 32 + j);
    p[jj] = a.P[idx];
    mk[jj] = gt_drop_mul(a.drop, dkey, idx);
    spd[i][j] = p[jj] * mk[jj];
  }
  for (int c0 = 0; c0 < a.hd; c0 += 32) {
    attn_load_slab(sa, a.dctx, a.lddc, b, h, a.hd, c0, tid, zp);
    attn_load_slab(sb, a.v, a.ldv, b, h, a.hd, c0, tid, zp);
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
      const float dv_ = sa[i][c];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) dp[jj] += dv_ * sb[jg + 8 * jj][c];
    }
    __syncthreads();
  }
  float rd = 0.f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) { dp[jj] *= mk[jj]; rd += dp[jj] * p[jj]; }
  rd += __shfl_xor(rd, 1); rd += __shfl_xor(rd, 2); rd += __shfl_xor(rd, 4);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) sds[i][jg + 8 * jj] = p[jj] * (dp[jj] - rd) * a.scale;
  __syncthreads();
  // here thread (i, jg) produces rows "i" of dq and rows "j = i" of dk / dv, columns jg + 8*cc
  for (int c0 = 0; c0 < a.hd; c0 += 32) {
    attn_load_slab(sa, a.dctx, a.lddc, b, h, a.hd, c0, tid, zp);
    attn_load_slab(sb, a.k, a.ldk, b, h, a.hd, c0, tid, zp);
    __syncthreads();
    float odv[4] = {0.f, 0.f, 0.f, 0.f}, odq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int t = 0; t < 32; ++t) {
      const float pd = spd[t][i];     // (P*mask)[t][j=i]
      const float ds = sds[i][t];     // dS[i][j=t]
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        odv[cc] += pd * sa[t][jg + 8 * cc];
        odq[cc] += ds * sb[t][jg + 8 * cc];
      }
    }
    __syncthreads();
    attn_load_slab(sa, a.q, a.ldq, b, h, a.hd, c0, tid, zp);
    __syncthreads();
    float odk[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int t = 0; t < 32; ++t) {
      const float ds = sds[t][i];     // dS[t][j=i]
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) odk[cc] += ds * sa[t][jg + 8 * cc];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = c0 + jg + 8 * cc;
      if (c < a.hd) {
        const size_t r = (size_t)(b * 32 + i);
        a.dq[r * a.lddq + h * a.hd + c] = odq[cc];
        a.dk[r * a.lddk + h * a.hd + c] = odk[cc];
        a.dv[r * a.lddv + h * a.hd + c] = odv[cc];
      }
    }
    __syncthreads();
  }
}


// ================================================================================================================
// MFMA variant (head_dim % 16 == 0).  Workgroup = 2 waves per (sequence, head); wave w owns query tile w (16 rows).
//
// The trick that removes every LDS round trip: compute the TRANSPOSED score tile  S^T = K Q^T  (A = K rows, B = Q rows).
// An MFMA result lives as D[row = 4g + r][col = l16]  (l16 = lane & 15, g = lane >> 4, r = register 0..3), so a lane
// holds  S^T[j = 16 tj + 4g + r][i = l16]  -- i.e. for ITS query row i = l16 the four keys 4g..4g+3 of each key tile.
//   * softmax over keys = in-lane over (tj, r), then over the 4 lane groups (xor 16, 32);
//   * as the A operand of the next product (A[m = l16][k = lane group g]) the same registers ARE  P[i][k]  for the
//     k-step numbering  k = 16 tj + 4g + c  (c = register): feeding V rows in that same order (B = V[16 tj + 4g + c][col])
//     makes  ctx = P V  a sequence of 8 MFMAs per 16 output columns with no data movement at all.
// Q/K fragments use the k-permutation of gt_gemm.h: lane (l16, g) loads ONE float4 = columns 16q + 4g .. +3 of its row
// and uses the components as four consecutive k-steps; A and B permute alike, so the contraction is unchanged.
// ================================================================================================================
// operand loads of the MFMA kernels: plain 16-byte / 4-byte loads, or (PAD: head_dim < 16 zero-padded to 16 columns) masked
// 4-byte loads through the zero page -- column index `col` (of the first element) against the real head_dim
template <bool PAD>
__device__ __forceinline__ float4 attn_ld4(const float* p, int col, int hd, const float* zp) {
  if (!PAD) return *reinterpret_cast<const float4*>(p);
  return make_float4(*(col < hd ? p : zp), *(col + 1 < hd ? p + 1 : zp), *(col + 2 < hd ? p + 2 : zp), *(col + 3 < hd ? p + 3 : zp));
}
template <bool PAD>
__device__ __forceinline__ float attn_ld1(const float* p, int col, int hd, const float* zp) {
  if (!PAD) return *p;
  return *(col < hd ? p : zp);
}
// (body: one (sequence, head) pair `bh`, query tile `ti` (0 / 1) per wave -- the stand-alone kernel runs it with 2 waves per
//  workgroup; gt_seq.h carries the same scheme on LDS operands for the sequence-resident kernels)
// operands of one (sequence, head): row 0 / first column of the head, in global memory or staged in LDS (the *_lds kernels)
struct AttnOps { const float* q; const float* k; const float* v; const float* dctx; int ldq, ldk, ldv, lddc; };
__device__ __forceinline__ AttnOps attn_ops_global(const AttnArgs& a, const int bh, const int hdr) {
  const size_t row0 = (size_t)(bh / a.H) * 32;
  const int hc = (bh % a.H) * hdr;
  return AttnOps{a.q + row0 * a.ldq + hc, a.k + row0 * a.ldk + hc, a.v + row0 * a.ldv + hc, a.dctx ? a.dctx + row0 * a.lddc + hc : nullptr, a.ldq, a.ldk, a.ldv, a.lddc};
}
template <int HD, bool PAD>
__device__ __forceinline__ void attn_fwd_mfma_body(const AttnArgs& a, const AttnOps& op, const int bh, const int ti, const int lane) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;               // real head_dim (PAD: < 16, operands zero-padded to 16 columns)
  const float* const zp = gt_zero_ptr();
  const int l16 = lane & 15, g = lane >> 4;
  const int b = bh / a.H, h = bh % a.H;
  const int i = 16 * ti + l16;                                   // this lane's query row
  const float* qrow = op.q + (size_t)i * op.ldq + 4 * g;
  const float* krow = op.k + (size_t)l16 * op.ldk + 4 * g;      // key tile 0; tile 1 = + 16 rows
  // Every operand of the kernel is requested up front (Q, K fragments and all of V: 40 registers at head_dim 32, 160 at
  // 128) so the loads are all in flight together; issued tile by tile each one would cost its own memory round trip.
  f32x4 st[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};           // S^T tiles [tj]
  float4 qf[NQ], k0[NQ], k1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    qf[q] = attn_ld4<PAD>(qrow + 16 * q, 16 * q + 4 * g, hdr, zp);
    k0[q] = attn_ld4<PAD>(krow + 16 * q, 16 * q + 4 * g, hdr, zp);
    k1[q] = attn_ld4<PAD>(krow + (size_t)16 * op.ldk + 16 * q, 16 * q + 4 * g, hdr, zp);
  }
  const float* __restrict__ vcol = op.v + (size_t)(4 * g) * op.ldv + l16;     // V[4g + c + 16 tj][16 ct + l16]
  float vb[NQ][2][4];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) vb[ct][tj][c] = attn_ld1<PAD>(vcol + (size_t)(16 * tj + c) * op.ldv + 16 * ct, 16 * ct + l16, hdr, zp);
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
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * tj + 4 * g + r;
      sv[tj][r] = (a.causal && j > i) ? -INFINITY : st[tj][r] * a.scale;
      mx = fmaxf(mx, sv[tj][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) { sv[tj][r] = expf(sv[tj][r] - mx); sum += sv[tj][r]; }
  sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  const uint32_t dkey = gt_drop_key(a.drop);
  float pd[2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const uint32_t idx0 = (uint32_t)((bh * 32 + i) * 32 + 16 * tj + 4 * g);
    float4 pv;
    pv.x = sv[tj][0] * inv; pv.y = sv[tj][1] * inv; pv.z = sv[tj][2] * inv; pv.w = sv[tj][3] * inv;
    *reinterpret_cast<float4*>(a.P + idx0) = pv;
    pd[tj][0] = pv.x * gt_drop_mul(a.drop, dkey, idx0);
    pd[tj][1] = pv.y * gt_drop_mul(a.drop, dkey, idx0 + 1);
    pd[tj][2] = pv.z * gt_drop_mul(a.drop, dkey, idx0 + 2);
    pd[tj][3] = pv.w * gt_drop_mul(a.drop, dkey, idx0 + 3);
  }
  // all V loads and MFMAs first, stores last: a store between two column tiles would pin the next tile's loads behind it
  // (the pointers may alias as far as the compiler knows) and every tile would pay a full memory round trip
  float* __restrict__ orow = a.ctx + (size_t)(b * 32 + 16 * ti + 4 * g) * a.ldc + h * hdr + l16;
  f32x4 o[NQ];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct) {
    o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) o[ct] = GT_MFMA16(pd[tj][c], vb[ct][tj][c], o[ct]);
  }
  if (a.ctx != nullptr) {                                   // (nullptr: ctx lives in bf16 only)
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { if (!PAD || 16 * ct + l16 < hdr) orow[(size_t)r * a.ldc + 16 * ct] = o[ct][r]; }
  }
  if (a.ctx16 != nullptr) {                                 // (dense (M, H hd) rows)
    uint16_t* __restrict__ o16 = a.ctx16 + (size_t)(b * 32 + 16 * ti + 4 * g) * (a.H * hdr) + h * hdr + l16;
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { if (!PAD || 16 * ct + l16 < hdr) o16[(size_t)r * (a.H * hdr) + 16 * ct] = gt_f2bf(o[ct][r]); }
  }
}

template <int HD, bool PAD>
__global__ __launch_bounds__(128) void attn_fwd_mfma_kernel(AttnArgs a) {
  attn_fwd_mfma_body<HD, PAD>(a, attn_ops_global(a, blockIdx.x, PAD ? a.hd : HD), blockIdx.x, threadIdx.x >> 6, threadIdx.x & 63);
}
// 32 x HD tile of one (sequence, head) into LDS (row stride HD + 4), from fp32 or -- IN16 -- from bf16 rows (8-byte loads of four elements,
// widened: a bf16 value is the upper half of its fp32 neighbour): every global byte requested once, all loads of a thread in flight together
template <int HD, int NT, bool IN16>
__device__ __forceinline__ void attn_stage_tile(float* dst, const float* src, const uint16_t* src16, const size_t row0, const int ld, const int hc, const int tid) {
  constexpr int LD = HD + 4, Q4 = HD / 4, PER = 32 * Q4 / NT;
  static_assert(32 * Q4 % NT == 0, "staging passes");
  float4 r[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = tid + NT * u, rr = e / Q4, c = (e % Q4) * 4;
    if (IN16) {
      const uint2 w = *reinterpret_cast<const uint2*>(src16 + (row0 + rr) * (size_t)ld + hc + c);
      r[u] = make_float4(gt_u2f(w.x << 16), gt_u2f(w.x & 0xFFFF0000u), gt_u2f(w.y << 16), gt_u2f(w.y & 0xFFFF0000u));
    } else {
      r[u] = *reinterpret_cast<const float4*>(src + (row0 + rr) * (size_t)ld + hc + c);
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = tid + NT * u;
    *reinterpret_cast<float4*>(dst + (e / Q4) * LD + (e % Q4) * 4) = r[u];
  }
}
// precision = 2: the forward with q / k / v staged in LDS from their bf16 storage (the register form above would read every bf16 row
// fragment 2-4 bytes at a time); the body is the same, on LDS operands
template <int HD, bool IN16>
__global__ __launch_bounds__(128) void attn_fwd_lds_kernel(AttnArgs a) {
  constexpr int LD = HD + 4;
  __shared__ __attribute__((aligned(16))) float sm[3 * 32 * LD];
  const int tid = threadIdx.x, bh = blockIdx.x;
  const size_t row0 = (size_t)(bh / a.H) * 32;
  const int hc = (bh % a.H) * HD;
  attn_stage_tile<HD, 128, IN16>(sm, a.q, a.q16, row0, a.ldq, hc, tid);
  attn_stage_tile<HD, 128, IN16>(sm + 32 * LD, a.k, a.k16, row0, a.ldk, hc, tid);
  attn_stage_tile<HD, 128, IN16>(sm + 64 * LD, a.v, a.v16, row0, a.ldv, hc, tid);
  __syncthreads();
  const AttnOps op{sm, sm + 32 * LD, sm + 64 * LD, nullptr, LD, LD, LD, LD};
  attn_fwd_mfma_body<HD, false>(a, op, bh, tid >> 6, tid & 63);
}

// Backward.  Each wave plays two roles, because dq contracts over keys and dk / dv contract over queries:
//   role 1 (query tile w, S^T layout as above): dPd^T = V dO^T, dS in registers -> dq rows of tile w;  row sums
//           rd[i] = sum_j dP P go through 32 floats of LDS so that role 2 can read the other wave's rows;
//   role 2 (key tile w, S layout: lane holds X[i = 16 ti + 4g + r][j = l16]): dPd = dO V^T, P reloaded in this layout,
//           (P*mask) and dS are then the A operands (A[m = j][k = i]) of  dv = (P*mask)^T dO  and  dk = dS^T q.
// (two bodies with a workgroup barrier between them -- srd, 32 floats of LDS per (sequence, head), carries the row sums)
template <int HD, bool PAD, int CS = 1>
__device__ __forceinline__ void attn_bwd_mfma_role1(const AttnArgs& a, const AttnOps& op, const int bh, const int w, const int lane, float* srd,
                                                    f32x4 (&dq_out)[HD / 16 / CS], const int cs = 0) {
  constexpr int NQ = HD / 16, NC = NQ / CS;                  // (CS: column split of attn_bwd_lds_kernel -- dP is every wave pair's, the dq column tiles are split)
  const int ct0 = cs * NC;
  const int hdr = PAD ? a.hd : HD;
  const float* const zp = gt_zero_ptr();
  const int l16 = lane & 15, g = lane >> 4;
  const uint32_t dkey = gt_drop_key(a.drop);

  // ---------------------------------------------------------------- role 1: query tile w
  {
    const int i = 16 * w + l16;
    const float* dorow = op.dctx + (size_t)i * op.lddc + 4 * g;
    const float* vrow = op.v + (size_t)l16 * op.ldv + 4 * g;
    f32x4 dt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};         // dPd^T tiles [tj]
    float4 df[NQ], v0[NQ], v1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {                       // all loads of this role first (see the forward kernel)
      df[q] = attn_ld4<PAD>(dorow + 16 * q, 16 * q + 4 * g, hdr, zp);
      v0[q] = attn_ld4<PAD>(vrow + 16 * q, 16 * q + 4 * g, hdr, zp);
      v1[q] = attn_ld4<PAD>(vrow + (size_t)16 * op.ldv + 16 * q, 16 * q + 4 * g, hdr, zp);
    }
    const float* __restrict__ kcol = op.k + (size_t)(4 * g) * op.ldk + l16;
    float kb[NC][2][4];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int c = 0; c < 4; ++c) kb[ct][tj][c] = attn_ld1<PAD>(kcol + (size_t)(16 * tj + c) * op.ldk + 16 * (ct0 + ct), 16 * (ct0 + ct) + l16, hdr, zp);
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
      const uint32_t idx0 = (uint32_t)((bh * 32 + i) * 32 + 16 * tj + 4 * g);
      const float4 pv = *reinterpret_cast<const float4*>(a.P + idx0);
      p[tj][0] = pv.x; p[tj][1] = pv.y; p[tj][2] = pv.z; p[tj][3] = pv.w;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dp[tj][r] = dt[tj][r] * gt_drop_mul(a.drop, dkey, idx0 + r);
        rd += dp[tj][r] * p[tj][r];
      }
    }
    rd += __shfl_xor(rd, 16); rd += __shfl_xor(rd, 32);
    if (g == 0 && (CS == 1 || cs == 0)) srd[i] = rd;
    float ds[2][4];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[tj][r] = p[tj][r] * (dp[tj][r] - rd) * a.scale;
    f32x4 o[NC];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int c = 0; c < 4; ++c) o[ct] = GT_MFMA16(ds[tj][c], kb[ct][tj][c], o[ct]);
    }
    // the dq stores wait until role 2 has issued its loads (dq may share a buffer with k / v: packed dqkv next to qkv is not
    // the case here, but the compiler cannot know) -- they are written at the very end of the kernel
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) dq_out[ct] = o[ct];
  }
}
// (dv / dk of key tile w are returned in registers: the caller stores them -- and role 1's dq -- see attn_bwd_store_direct)
template <int HD, bool PAD, int CS = 1>
__device__ __forceinline__ void attn_bwd_mfma_role2(const AttnArgs& a, const AttnOps& op, const int bh, const int w, const int lane, const float* srd,
                                                    f32x4 (&ov)[HD / 16 / CS], f32x4 (&ok)[HD / 16 / CS], const int cs = 0) {
  constexpr int NQ = HD / 16, NC = NQ / CS;
  const int ct0 = cs * NC;
  const int hdr = PAD ? a.hd : HD;
  const float* const zp = gt_zero_ptr();
  const int l16 = lane & 15, g = lane >> 4;
  const uint32_t dkey = gt_drop_key(a.drop);
  // ---------------------------------------------------------------- role 2: key tile w
  {
    const int j = 16 * w + l16;
    const float* dorow = op.dctx + (size_t)l16 * op.lddc + 4 * g;               // query tile 0; tile 1 = + 16 rows
    const float* vrow = op.v + (size_t)j * op.ldv + 4 * g;
    f32x4 dd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};         // dPd tiles [ti]
    float4 vf[NQ], d0[NQ], d1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      vf[q] = attn_ld4<PAD>(vrow + 16 * q, 16 * q + 4 * g, hdr, zp);
      d0[q] = attn_ld4<PAD>(dorow + 16 * q, 16 * q + 4 * g, hdr, zp);
      d1[q] = attn_ld4<PAD>(dorow + (size_t)16 * op.lddc + 16 * q, 16 * q + 4 * g, hdr, zp);
    }
    const float* __restrict__ docol = op.dctx + (size_t)(4 * g) * op.lddc + l16;
    const float* __restrict__ qcol = op.q + (size_t)(4 * g) * op.ldq + l16;
    float db[NC][2][4], qb[NC][2][4];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          db[ct][ti][c] = attn_ld1<PAD>(docol + (size_t)(16 * ti + c) * op.lddc + 16 * (ct0 + ct), 16 * (ct0 + ct) + l16, hdr, zp);
          qb[ct][ti][c] = attn_ld1<PAD>(qcol + (size_t)(16 * ti + c) * op.ldq + 16 * (ct0 + ct), 16 * (ct0 + ct) + l16, hdr, zp);
        }
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
        const uint32_t idx = (uint32_t)((bh * 32 + i) * 32 + j);
        const float pv = a.P[idx];
        const float mk = gt_drop_mul(a.drop, dkey, idx);
        pdm[ti][r] = pv * mk;
        ds[ti][r] = pv * (dd[ti][r] * mk - srd[i]) * a.scale;
      }
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      ov[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; ok[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ov[ct] = GT_MFMA16(pdm[ti][c], db[ct][ti][c], ov[ct]);
          ok[ct] = GT_MFMA16(ds[ti][c], qb[ct][ti][c], ok[ct]);
        }
    }
  }
}
// rows 16 w + 4 g + r, columns 16 ct + l16 of the head, straight from the accumulator registers (64-byte segments)
template <int HD, bool PAD>
__device__ __forceinline__ void attn_bwd_store_direct(const AttnArgs& a, const int bh, const int w, const int lane, const f32x4 (&dq_out)[HD / 16],
                                                      const f32x4 (&ov)[HD / 16], const f32x4 (&ok)[HD / 16]) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)(bh / a.H) * 32;
  const int hc = (bh % a.H) * hdr;
  float* __restrict__ dqrow = a.dq + (row0 + 16 * w + 4 * g) * a.lddq + hc + l16;
  float* __restrict__ dvrow = a.dv + (row0 + 16 * w + 4 * g) * a.lddv + hc + l16;
  float* __restrict__ dkrow = a.dk + (row0 + 16 * w + 4 * g) * a.lddk + hc + l16;
  if (a.dq != nullptr) {
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (PAD && 16 * ct + l16 >= hdr) continue;
        dqrow[(size_t)r * a.lddq + 16 * ct] = dq_out[ct][r];
        dvrow[(size_t)r * a.lddv + 16 * ct] = ov[ct][r];
        dkrow[(size_t)r * a.lddk + 16 * ct] = ok[ct][r];
      }
  }
  if (a.dqkv16 != nullptr) {                                // [dq | dk | dv], dense (M, 3 d) rows
    const int d = a.H * hdr;
    uint16_t* __restrict__ o16 = a.dqkv16 + (row0 + 16 * w + 4 * g) * (size_t)(3 * d) + hc + l16;
#pragma unroll
    for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (PAD && 16 * ct + l16 >= hdr) continue;
        o16[(size_t)r * (3 * d) + 16 * ct] = gt_f2bf(dq_out[ct][r]);
        o16[(size_t)r * (3 * d) + 16 * ct + d] = gt_f2bf(ok[ct][r]);
        o16[(size_t)r * (3 * d) + 16 * ct + 2 * d] = gt_f2bf(ov[ct][r]);
      }
  }
}
template <int HD, bool PAD>
__global__ __launch_bounds__(128) void attn_bwd_mfma_kernel(AttnArgs a) {
  __shared__ float srd[32];
  f32x4 dq_out[HD / 16], ov[HD / 16], ok[HD / 16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const AttnOps op = attn_ops_global(a, blockIdx.x, PAD ? a.hd : HD);
  attn_bwd_mfma_role1<HD, PAD>(a, op, blockIdx.x, w, lane, srd, dq_out);
  __syncthreads();
  attn_bwd_mfma_role2<HD, PAD>(a, op, blockIdx.x, w, lane, srd, ov, ok);
  attn_bwd_store_direct<HD, PAD>(a, blockIdx.x, w, lane, dq_out, ov, ok);
}

// The same backward with the four operand tiles of the (sequence, head) -- q, k, v, dctx: 32 x HD each -- staged in LDS first: every
// global byte is requested ONCE, by 16-byte loads that are all in flight together (the register form above reads each tile two or three
// times, the column-major fragments 4 bytes at a time: L2 absorbs that, but at d_model 512 it held the kernel to 3.2 TB/s), the
// fragments then come from LDS (row stride HD + 4: the 16 rows of a fragment read fall in different banks), and dq / dk / dv leave
// through the same LDS tiles as full 256-byte row segments.
// CS > 1 (few (sequence, head) pairs, wide heads): 2 CS waves per pair -- every wave pair repeats the dP contraction of its query / key tile
// from LDS and takes 1 / CS of the head's column tiles (role bodies above); the staging loads and the row stores spread over all of them.
template <int HD, int CS = 1, bool IN16 = false>
__global__ __launch_bounds__(128 * CS) void attn_bwd_lds_kernel(AttnArgs a) {
  constexpr int NT = 128 * CS, LD = HD + 4, Q4 = HD / 4, PER = 32 * Q4 / NT, NC = HD / 16 / CS;
  static_assert(32 * Q4 % NT == 0, "staging passes");
  __shared__ __attribute__((aligned(16))) float sm[4 * 32 * LD];
  __shared__ float srd[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = wave & 1, cs = wave >> 1, bh = blockIdx.x;
  if constexpr (IN16) {                                          // precision = 2: q / k / v / dctx stored in bf16 alone
    const size_t r0 = (size_t)(bh / a.H) * 32;
    const int hc0 = (bh % a.H) * HD;
    attn_stage_tile<HD, NT, true>(sm, nullptr, a.q16, r0, a.ldq, hc0, tid);
    attn_stage_tile<HD, NT, true>(sm + 32 * LD, nullptr, a.k16, r0, a.ldk, hc0, tid);
    attn_stage_tile<HD, NT, true>(sm + 64 * LD, nullptr, a.v16, r0, a.ldv, hc0, tid);
    attn_stage_tile<HD, NT, true>(sm + 96 * LD, nullptr, a.dctx16, r0, a.lddc, hc0, tid);
  } else {
  const AttnOps og = attn_ops_global(a, bh, HD);
  float4 rq[PER], rk[PER], rv[PER], rd[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = tid + NT * u, r = e / Q4, c = (e % Q4) * 4;
    rq[u] = *reinterpret_cast<const float4*>(og.q + (size_t)r * og.ldq + c);
    rk[u] = *reinterpret_cast<const float4*>(og.k + (size_t)r * og.ldk + c);
    rv[u] = *reinterpret_cast<const float4*>(og.v + (size_t)r * og.ldv + c);
    rd[u] = *reinterpret_cast<const float4*>(og.dctx + (size_t)r * og.lddc + c);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = tid + NT * u, o = (e / Q4) * LD + (e % Q4) * 4;
    *reinterpret_cast<float4*>(sm + o) = rq[u];
    *reinterpret_cast<float4*>(sm + 32 * LD + o) = rk[u];
    *reinterpret_cast<float4*>(sm + 64 * LD + o) = rv[u];
    *reinterpret_cast<float4*>(sm + 96 * LD + o) = rd[u];
  }
  }
  __syncthreads();
  const AttnOps op{sm, sm + 32 * LD, sm + 64 * LD, sm + 96 * LD, LD, LD, LD, LD};
  f32x4 dq_out[NC], ov[NC], ok[NC];
  attn_bwd_mfma_role1<HD, false, CS>(a, op, bh, w, lane, srd, dq_out, cs);
  __syncthreads();
  attn_bwd_mfma_role2<HD, false, CS>(a, op, bh, w, lane, srd, ov, ok, cs);
  __syncthreads();                                               // every fragment read of q / k / v is done: the tiles take dq / dk / dv
  {
    const int l16 = lane & 15, g = lane >> 4;
    float* o = sm + (16 * w + 4 * g) * LD + 16 * cs * NC + l16;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r * LD + 16 * ct] = dq_out[ct][r];
        o[32 * LD + r * LD + 16 * ct] = ok[ct][r];
        o[64 * LD + r * LD + 16 * ct] = ov[ct][r];
      }
  }
  __syncthreads();
  const size_t row0 = (size_t)(bh / a.H) * 32;
  const int hc = (bh % a.H) * HD;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = tid + NT * u, r = e / Q4, c = (e % Q4) * 4, o = r * LD + c;
    if (a.dq != nullptr) {                                  // (nullptr: dq / dk / dv live in bf16 only)
      *reinterpret_cast<float4*>(a.dq + (row0 + r) * a.lddq + hc + c) = *reinterpret_cast<const float4*>(sm + o);
      *reinterpret_cast<float4*>(a.dk + (row0 + r) * a.lddk + hc + c) = *reinterpret_cast<const float4*>(sm + 32 * LD + o);
      *reinterpret_cast<float4*>(a.dv + (row0 + r) * a.lddv + hc + c) = *reinterpret_cast<const float4*>(sm + 64 * LD + o);
    }
    if (a.dqkv16 != nullptr) {                              // [dq | dk | dv], dense (M, 3 d) rows
      const int d = a.H * HD;
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        const float4 v = *reinterpret_cast<const float4*>(sm + part * 32 * LD + o);
        uint2 pk;
        pk.x = (uint32_t)gt_f2bf(v.x) | ((uint32_t)gt_f2bf(v.y) << 16); pk.y = (uint32_t)gt_f2bf(v.z) | ((uint32_t)gt_f2bf(v.w) << 16);
        *reinterpret_cast<uint2*>(a.dqkv16 + (row0 + r) * (size_t)(3 * d) + part * d + hc + c) = pk;
      }
    }
  }
}


// ================================================================================================================
// Greedy decoding (gt_predict, encoder-decoder): ONE query row per (sequence, head) against the first `nkeys` key/value
// rows of the sequence -- the rows the earlier decode steps wrote are the KV cache (self-attention: nkeys = t + 1, which
// IS the causal mask), or the cross-attention keys/values of the encoder memory (nkeys = 32).  No dropout (eval), P is
// not saved.  One wave per (sequence, head): lane j scores key j, wave reductions for the softmax, then lanes own output
// columns.  q / ctx point at the query's time step; row stride between sequences is 32 * ld.
// ================================================================================================================
__global__ __launch_bounds__(64) void attn_decode_kernel(AttnArgs a, int nkeys) {
  // K is walked in 64-column slabs: 32 coalesced row loads (one 256-byte row segment per instruction, all in flight
  // together) staged in LDS, from which lane j takes the dot product of key j.  (A first version let every lane walk its
  // own key row straight from global memory: 32 cache lines per load instruction, 265 us per call at 8192 heads.)
  __shared__ float sk[32][65], sq[64], sp[32];
  const int lane = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh % a.H, hd = a.hd;
  const float* const zp = gt_zero_ptr();
  const float* q = a.q + (size_t)b * 32 * a.ldq + h * hd;
  const float* kb = a.k + (size_t)b * 32 * a.ldk + h * hd;
  const float* vb = a.v + (size_t)b * 32 * a.ldv + h * hd;
  float s = 0.f;
  for (int c0 = 0; c0 < hd; c0 += 64) {
    const bool okc = c0 + lane < hd;
    float kr[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) kr[j] = *((okc && j < nkeys) ? kb + (size_t)j * a.ldk + c0 + lane : zp);
    sq[lane] = *(okc ? q + c0 + lane : zp);
#pragma unroll
    for (int j = 0; j < 32; ++j) sk[j][lane] = kr[j];
    __syncthreads();
    if (lane < 32) {
#pragma unroll 16
      for (int c = 0; c < 64; ++c) s += sq[c] * sk[lane][c];
    }
    __syncthreads();
  }
  s = (lane < nkeys) ? s * a.scale : -INFINITY;
  float mx = s;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
  float e = (lane < nkeys) ? expf(s - mx) : 0.f, sum = e;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
  if (lane < 32) sp[lane] = e / sum;
  __syncthreads();
  float* o = a.ctx + (size_t)b * 32 * a.ldc + h * hd;
  for (int c0 = 0; c0 < hd; c0 += 64) {
    const bool okc = c0 + lane < hd;
    float vr[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) vr[j] = *((okc && j < nkeys) ? vb + (size_t)j * a.ldv + c0 + lane : zp);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) acc += sp[j] * vr[j];          // sp[j] = 0 beyond nkeys
    if (okc) o[c0 + lane] = acc;
  }
}
