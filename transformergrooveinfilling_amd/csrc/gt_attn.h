// T=32 multi-head attention core, forward and backward, one workgroup per (sequence, head).
//
// Follows torch:nn/functional.py:6504-6642: S = (q k^T)/sqrt(hd) [+ causal -inf], P = softmax(S),
// dropout(P), ctx = P v.  A whole 32x32 score tile lives in registers/LDS of one workgroup (8 lanes
// per query row), so there is no online softmax and nothing to shard.  The attention core is 2-5 %
// of the step's FLOPs; it is LDS-tiled VALU code, generic in head_dim (1..512) by
// walking head_dim in 32-column slabs.
// Saved for backward: P (pre-dropout probabilities, (B*H,32,32)); the dropout mask is regenerated
// from the counter-based hash.
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
