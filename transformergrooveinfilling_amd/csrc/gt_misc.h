// Row-wise / elementwise kernels of the step: standalone LayerNorm fwd/bwd (final encoder/decoder
// norms), the fused BCE+MSE loss with its hit_loss_penalty mask, head-activation backward,
// the flat multi-tensor SGD/Adam update, teacher-forcing shift and the predict threshold.
// All are HBM-bound streaming kernels; every (M,d) pass is one coalesced read + one write.
#pragma once
#include "gt_common.h"

// ---- LayerNorm forward: one wave per row, the row lives in registers --------------------------------
//   z = x * dropmask(+ res);  y = LN(z) * gamma + beta;  xhat, rstd saved for backward.   y may alias x (in place).
//   x / res / y have their own leading dimensions (the greedy decoder works on one time step of every sequence: ld = 32 N);
//   xhat / rstd are dense (M, N) / (M).  The dropout index is row * N + c (dense rows: training never uses a strided view).
// Used for the final encoder / decoder norms and -- with res / dropout -- as the second half of the UN-fused
// `LN(drop(linear) + res)` (wide d_model at few tokens, where a row-owning GEMM tile would make every workgroup stream
// the whole weight matrix; see linear_res_ln in groove_hip.hip).
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, const float* __restrict__ res, DropArgs drop,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* y,
                                                     float* __restrict__ xhat, float* __restrict__ rstd_out, int M, int N,
                                                     int ldx, int ldres, int ldy, uint16_t* __restrict__ y16 = nullptr,
                                                     const uint16_t* x16 = nullptr) {
  // y16: a bf16 copy of y (dense rows of N), for the GEMM that takes y as its operand at precision = 1 (GemmArgs::A16)
  // x16 (precision = 2): the Linear output ahead of this norm, stored in bf16 alone (dense rows of N; may be the y16 region: a row is read
  // into registers before any of it is written)
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* const zp = gt_zero_ptr();
  const uint32_t dkey = gt_drop_key(drop);
  const float invN = 1.0f / (float)N;
  float z[GT_MAX_D / 64], r[GT_MAX_D / 64], ga[GT_MAX_D / 64], be[GT_MAX_D / 64];
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {            // all loads first, branch-free (address select)
    const int c = lane + 64 * i;
    const bool ok = c < N;
    r[i] = *((ok && res != nullptr) ? res + (size_t)row * ldres + c : zp);
    ga[i] = *(ok ? gamma + c : zp);
    be[i] = *(ok ? beta + c : zp);
  }
  if (x16 != nullptr) {                                // (one uniform branch around ALL the loads: a select per element would serialise them)
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      z[i] = gt_bf2f(*(c < N ? x16 + (size_t)row * N + c : reinterpret_cast<const uint16_t*>(zp)));
    }
  } else {
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      z[i] = *(c < N ? x + (size_t)row * ldx + c : zp);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {
    const int c = lane + 64 * i;
    z[i] = (c < N) ? z[i] * gt_drop_mul(drop, dkey, (uint32_t)((size_t)row * N + c)) + r[i] : 0.f;
    s += z[i];
  }
  const float mean = gt_wave_sum(s) * invN;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) { if (lane + 64 * i < N) { const float d = z[i] - mean; q += d * d; } }
  const float rstd = 1.0f / sqrtf(gt_wave_sum(q) * invN + GT_LN_EPS);
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {
    const int c = lane + 64 * i;
    if (c < N) {
      const float xh = (z[i] - mean) * rstd;
      xhat[(size_t)row * N + c] = xh;
      const float yv = xh * ga[i] + be[i];
      y[(size_t)row * ldy + c] = yv;
      if (y16 != nullptr) y16[(size_t)row * N + c] = gt_f2bf(yv);
    }
  }
  if (lane == 0) rstd_out[row] = rstd;
}

// ---- LayerNorm backward: one wave walks `rows_per_wave` rows; dgamma/dbeta leave as one partial per workgroup ------
//   g = dy (+ res);  dz = LNbwd(g);  dz_masked = dz * dropmask.   dz may alias dy (in place).
#define GT_LNB_ROWS 8
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* __restrict__ res, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* dz, float* __restrict__ dz_masked, DropArgs drop,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ part,
                                                     int M, int N, int rows_per_wave, uint16_t* __restrict__ dzm16 = nullptr) {
  __shared__ float sred[4][2][GT_MAX_D];
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave;
  const float invN = 1.0f / (float)N;
  const uint32_t dkey = gt_drop_key(drop);
  const float* const zp = gt_zero_ptr();
  float dg[GT_MAX_D / 64], db[GT_MAX_D / 64];
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) { dg[i] = 0.f; db[i] = 0.f; }
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    float s1 = 0.f, s2 = 0.f, d[GT_MAX_D / 64], e[GT_MAX_D / 64], xh[GT_MAX_D / 64], ga[GT_MAX_D / 64];
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {          // all loads first, branch-free (address select)
      const int c = lane + 64 * i;
      const bool ok = c < N;
      d[i] = *(ok ? dy + (size_t)row * N + c : zp);
      e[i] = *((ok && res != nullptr) ? res + (size_t)row * N + c : zp);
      xh[i] = *(ok ? xhat + (size_t)row * N + c : zp);
      ga[i] = *(ok ? gamma + c : zp);
    }
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      d[i] += e[i];
      const float gd = d[i] * ga[i];
      s1 += gd; s2 += gd * xh[i]; dg[i] += d[i] * xh[i]; db[i] += d[i];
    }
    const float m1 = gt_wave_sum(s1) * invN, m2 = gt_wave_sum(s2) * invN, rs = rstd[row];
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      if (c < N) {
        const size_t e = (size_t)row * N + c;
        const float v = rs * (d[i] * ga[i] - m1 - xh[i] * m2);
        dz[e] = v;
        const float vm = (dz_masked || (dzm16 && drop.thr != 0u && drop.st != nullptr)) ? v * gt_drop_mul(drop, dkey, (uint32_t)e) : v;
        if (dz_masked) dz_masked[e] = vm;
        if (dzm16) dzm16[e] = gt_f2bf(vm);
      }
    }
  }
  if (part == nullptr) {
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      if (c < N && row0 < M) { atomicAdd(&dgamma[c], dg[i]); atomicAdd(&dbeta[c], db[i]); }
    }
    return;
  }
  // partials per workgroup [block][2][N] (summed by ln_param_reduce_kernel): no contended atomics
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {
    const int c = lane + 64 * i;
    if (c < N) { sred[w][0][c] = dg[i]; sred[w][1][c] = db[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < N; c += 256) {
    part[((size_t)blockIdx.x * 2) * N + c] = sred[0][0][c] + sred[1][0][c] + sred[2][0][c] + sred[3][0][c];
    part[((size_t)blockIdx.x * 2 + 1) * N + c] = sred[0][1][c] + sred[1][1][c] + sred[2][1][c] + sred[3][1][c];
  }
}

// The same for rows of 256 / 512 floats (d_model 256 / 512; NV = N / 256): 16-byte accesses (a wave instruction moves 1 KB instead of
// 256 B), gamma fetched once per wave instead of once per row.  Round 3: at d_model 512 the row passes are 9 % (fp32) / 19 % (bf16
// operands) of the step and this one ran at 3.7 TB/s of HBM-side traffic.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_v4_kernel(const float* dy, const float* __restrict__ res, const float* __restrict__ xhat,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        float* dz, float* __restrict__ dz_masked, DropArgs drop,
                                                        float* __restrict__ part, int M, int rows_per_wave, uint16_t* __restrict__ dzm16 = nullptr,
                                                        const uint16_t* dy16 = nullptr) {
  // dzm16: a bf16 copy of the tensor the next dgrad / weight gradient takes as its operand (dz_masked, or dz when there is no mask)
  // dy16 (precision = 2): the dgrad output ahead of this backward, stored in bf16 alone (may be the dzm16 region: read before written)
  constexpr int N = 256 * NV;
  __shared__ float sred[4][2][N];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + w) * rows_per_wave;
  const float invN = 1.0f / (float)N;
  const uint32_t dkey = gt_drop_key(drop);
  float4 ga[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ga[i] = *reinterpret_cast<const float4*>(gamma + 4 * lane + 256 * i);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    const size_t base = (size_t)row * N + 4 * lane;
    float4 d[NV], xh[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) xh[i] = *reinterpret_cast<const float4*>(xhat + base + 256 * i);      // all loads first
    if (dy16 != nullptr) {                             // (one uniform branch around the loads of d)
      uint2 w2[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) w2[i] = *reinterpret_cast<const uint2*>(dy16 + base + 256 * i);
#pragma unroll
      for (int i = 0; i < NV; ++i) d[i] = make_float4(gt_u2f(w2[i].x << 16), gt_u2f(w2[i].x & 0xFFFF0000u), gt_u2f(w2[i].y << 16), gt_u2f(w2[i].y & 0xFFFF0000u));
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) d[i] = *reinterpret_cast<const float4*>(dy + base + 256 * i);
    }
    if (res != nullptr) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 e = *reinterpret_cast<const float4*>(res + base + 256 * i);
        d[i].x += e.x; d[i].y += e.y; d[i].z += e.z; d[i].w += e.w;
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float gx = d[i].x * ga[i].x, gy = d[i].y * ga[i].y, gz = d[i].z * ga[i].z, gw = d[i].w * ga[i].w;
      s1 += (gx + gy) + (gz + gw);
      s2 += (gx * xh[i].x + gy * xh[i].y) + (gz * xh[i].z + gw * xh[i].w);
      dg[i].x += d[i].x * xh[i].x; dg[i].y += d[i].y * xh[i].y; dg[i].z += d[i].z * xh[i].z; dg[i].w += d[i].w * xh[i].w;
      db[i].x += d[i].x; db[i].y += d[i].y; db[i].z += d[i].z; db[i].w += d[i].w;
    }
    const float m1 = gt_wave_sum(s1) * invN, m2 = gt_wave_sum(s2) * invN, rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 v;
      v.x = rs * (d[i].x * ga[i].x - m1 - xh[i].x * m2); v.y = rs * (d[i].y * ga[i].y - m1 - xh[i].y * m2);
      v.z = rs * (d[i].z * ga[i].z - m1 - xh[i].z * m2); v.w = rs * (d[i].w * ga[i].w - m1 - xh[i].w * m2);
      *reinterpret_cast<float4*>(dz + base + 256 * i) = v;
      float4 vm = v;
      if (dz_masked != nullptr || (dzm16 != nullptr && drop.thr != 0u && drop.st != nullptr)) {      // (the masked copy may live in bf16 only)
        const uint32_t e = (uint32_t)(base + 256 * i);
        vm = make_float4(v.x * gt_drop_mul(drop, dkey, e), v.y * gt_drop_mul(drop, dkey, e + 1), v.z * gt_drop_mul(drop, dkey, e + 2),
                         v.w * gt_drop_mul(drop, dkey, e + 3));
        if (dz_masked != nullptr) *reinterpret_cast<float4*>(dz_masked + base + 256 * i) = vm;
      }
      if (dzm16 != nullptr) {
        uint2 pk;
        pk.x = (uint32_t)gt_f2bf(vm.x) | ((uint32_t)gt_f2bf(vm.y) << 16); pk.y = (uint32_t)gt_f2bf(vm.z) | ((uint32_t)gt_f2bf(vm.w) << 16);
        *reinterpret_cast<uint2*>(dzm16 + base + 256 * i) = pk;
      }
    }
  }
  // partials per workgroup [block][2][N] (summed by ln_param_reduce_kernel)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    *reinterpret_cast<float4*>(&sred[w][0][4 * lane + 256 * i]) = dg[i];
    *reinterpret_cast<float4*>(&sred[w][1][4 * lane + 256 * i]) = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < N; c += 256) {
    part[((size_t)blockIdx.x * 2) * N + c] = sred[0][0][c] + sred[1][0][c] + sred[2][0][c] + sred[3][0][c];
    part[((size_t)blockIdx.x * 2 + 1) * N + c] = sred[0][1][c] + sred[1][1][c] + sred[2][1][c] + sred[3][1][c];
  }
}

// ---- two LayerNorms back to back in one pass (the top layer's last norm and the final encoder / decoder norm) -----------
// forward:  z = x * dropmask + res;  y1 = LN_1(z);  y2 = LN_2(y1)      (xhat / rstd of both saved; all outputs dense (M, N))
__global__ __launch_bounds__(256) void ln_fwd2_kernel(const float* x, const float* __restrict__ res, DropArgs drop,
                                                      const float* __restrict__ gamma1, const float* __restrict__ beta1, float* y1,
                                                      float* __restrict__ xhat1, float* __restrict__ rstd1,
                                                      const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                      float* __restrict__ y2, float* __restrict__ xhat2, float* __restrict__ rstd2,
                                                      int M, int N, const uint16_t* x16 = nullptr) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* const zp = gt_zero_ptr();
  const uint32_t dkey = gt_drop_key(drop);
  const float invN = 1.0f / (float)N;
  float z[GT_MAX_D / 64], r[GT_MAX_D / 64], ga[GT_MAX_D / 64], be[GT_MAX_D / 64], gb[GT_MAX_D / 64], bb[GT_MAX_D / 64];
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {            // all loads first, branch-free (address select)
    const int c = lane + 64 * i;
    const bool ok = c < N;
    r[i] = *((ok && res != nullptr) ? res + (size_t)row * N + c : zp);
    ga[i] = *(ok ? gamma1 + c : zp); be[i] = *(ok ? beta1 + c : zp);
    gb[i] = *(ok ? gamma2 + c : zp); bb[i] = *(ok ? beta2 + c : zp);
  }
  if (x16 != nullptr) {
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      z[i] = gt_bf2f(*(c < N ? x16 + (size_t)row * N + c : reinterpret_cast<const uint16_t*>(zp)));
    }
  } else {
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      z[i] = *(c < N ? x + (size_t)row * N + c : zp);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {
    const int c = lane + 64 * i;
    z[i] = (c < N) ? z[i] * gt_drop_mul(drop, dkey, (uint32_t)((size_t)row * N + c)) + r[i] : 0.f;
    s += z[i];
  }
  float mean = gt_wave_sum(s) * invN, q = 0.f;
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) { if (lane + 64 * i < N) { const float d = z[i] - mean; q += d * d; } }
  float rs = 1.0f / sqrtf(gt_wave_sum(q) * invN + GT_LN_EPS);
  s = 0.f;
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {
    const int c = lane + 64 * i;
    if (c < N) {
      const float xh = (z[i] - mean) * rs;
      xhat1[(size_t)row * N + c] = xh;
      z[i] = xh * ga[i] + be[i];
      y1[(size_t)row * N + c] = z[i];
      s += z[i];
    }
  }
  if (lane == 0) rstd1[row] = rs;
  mean = gt_wave_sum(s) * invN; q = 0.f;
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) { if (lane + 64 * i < N) { const float d = z[i] - mean; q += d * d; } }
  rs = 1.0f / sqrtf(gt_wave_sum(q) * invN + GT_LN_EPS);
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) {
    const int c = lane + 64 * i;
    if (c < N) {
      const float xh = (z[i] - mean) * rs;
      xhat2[(size_t)row * N + c] = xh;
      y2[(size_t)row * N + c] = xh * gb[i] + bb[i];
    }
  }
  if (lane == 0) rstd2[row] = rs;
}

// backward:  g = dy;  t = LNbwd_2(g)  (the OUTER norm, applied last in forward);  dz = LNbwd_1(t);  dz_masked = dz * dropmask.
// Both norms leave their dgamma / dbeta as per-workgroup partials (part2 for the outer norm, part1 for the inner one).
__global__ __launch_bounds__(256) void ln_bwd2_kernel(const float* dy, const float* __restrict__ xhat2, const float* __restrict__ rstd2,
                                                      const float* __restrict__ gamma2, float* __restrict__ part2,
                                                      const float* __restrict__ xhat1, const float* __restrict__ rstd1,
                                                      const float* __restrict__ gamma1, float* __restrict__ part1,
                                                      float* dz, float* __restrict__ dz_masked, DropArgs drop, int M, int N,
                                                      int rows_per_wave, uint16_t* __restrict__ dzm16 = nullptr) {
  __shared__ float sred[4][2][GT_MAX_D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + w) * rows_per_wave;
  const float invN = 1.0f / (float)N;
  const uint32_t dkey = gt_drop_key(drop);
  const float* const zp = gt_zero_ptr();
  float dg2[GT_MAX_D / 64], db2[GT_MAX_D / 64], dg1[GT_MAX_D / 64], db1[GT_MAX_D / 64];
#pragma unroll
  for (int i = 0; i < GT_MAX_D / 64; ++i) { dg2[i] = 0.f; db2[i] = 0.f; dg1[i] = 0.f; db1[i] = 0.f; }
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    float d[GT_MAX_D / 64], xa[GT_MAX_D / 64], gA[GT_MAX_D / 64], xb[GT_MAX_D / 64], gB[GT_MAX_D / 64];
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {          // all loads first, branch-free (address select)
      const int c = lane + 64 * i;
      const bool ok = c < N;
      d[i] = *(ok ? dy + (size_t)row * N + c : zp);
      xa[i] = *(ok ? xhat2 + (size_t)row * N + c : zp); gA[i] = *(ok ? gamma2 + c : zp);
      xb[i] = *(ok ? xhat1 + (size_t)row * N + c : zp); gB[i] = *(ok ? gamma1 + c : zp);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const float gd = d[i] * gA[i];
      s1 += gd; s2 += gd * xa[i]; dg2[i] += d[i] * xa[i]; db2[i] += d[i];
    }
    float m1 = gt_wave_sum(s1) * invN, m2 = gt_wave_sum(s2) * invN, rs = rstd2[row];
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      d[i] = (lane + 64 * i < N) ? rs * (d[i] * gA[i] - m1 - xa[i] * m2) : 0.f;        // gradient w.r.t. the inner norm's output
      const float gd = d[i] * gB[i];
      s1 += gd; s2 += gd * xb[i]; dg1[i] += d[i] * xb[i]; db1[i] += d[i];
    }
    m1 = gt_wave_sum(s1) * invN; m2 = gt_wave_sum(s2) * invN; rs = rstd1[row];
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      if (c < N) {
        const size_t e = (size_t)row * N + c;
        const float v = rs * (d[i] * gB[i] - m1 - xb[i] * m2);
        dz[e] = v;
        const float vm = (dz_masked || (dzm16 && drop.thr != 0u && drop.st != nullptr)) ? v * gt_drop_mul(drop, dkey, (uint32_t)e) : v;
        if (dz_masked) dz_masked[e] = vm;
        if (dzm16) dzm16[e] = gt_f2bf(vm);
      }
    }
  }
  // partials per workgroup [block][2][N], outer norm then inner norm (the LDS buffer is reused)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* part = pass == 0 ? part2 : part1;
#pragma unroll
    for (int i = 0; i < GT_MAX_D / 64; ++i) {
      const int c = lane + 64 * i;
      if (c < N) { sred[w][0][c] = pass == 0 ? dg2[i] : dg1[i]; sred[w][1][c] = pass == 0 ? db2[i] : db1[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
      part[((size_t)blockIdx.x * 2) * N + c] = sred[0][0][c] + sred[1][0][c] + sred[2][0][c] + sred[3][0][c];
      part[((size_t)blockIdx.x * 2 + 1) * N + c] = sred[0][1][c] + sred[1][1][c] + sred[2][1][c] + sred[3][1][c];
    }
    __syncthreads();
  }
}

// dgamma / dbeta of every LayerNorm of the step in ONE launch: job j sums its workgroup partials [nwg][2][N] in a fixed
// order (deterministic) and adds them into the gradient buffer.  grid = (ceil(2N/64), jobs); 16 waves split the partials.
#define GT_LN_JOBS_MAX 96
struct LnJob { const float* part; float* dgamma; float* dbeta; int nwg; };
struct LnJobs {
  int n, N;
  gt_step_state* bump;       // fused train step: this launch (the last of backward; nothing after it regenerates a dropout mask) also
                             // advances step / opt_step -- the optimizer that follows is told so -- saving the step_inc launch
  unsigned* err;             // ... looking at the exchange region's error word (nullptr: none) as the update will: gt_bump_counters
  LnJob j[GT_LN_JOBS_MAX];
};
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(LnJobs jobs) {
  // 16 waves split the partial rows; every lane has 8 independent loads in flight per trip (a 4-wave version walking 16
  // dependent trips took 8.4 us for 3.7 MB of partials: latency, not bandwidth).  Fixed summation order -> deterministic.
  __shared__ float s[16][64];
  if (jobs.bump != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) gt_bump_counters(jobs.bump, jobs.err, nullptr);
  const LnJob jb = jobs.j[blockIdx.y];
  const float* const zp = gt_zero_ptr();
  const int N = jobs.N, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c2 = blockIdx.x * 64 + lane;                 // column in [0, 2N): gamma then beta
  const bool ok = c2 < 2 * N;
  const int which = (ok && c2 >= N) ? 1 : 0, c = ok ? c2 - which * N : 0;
  float acc = 0.f;
  for (int g0 = w; g0 < jb.nwg; g0 += 128) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gidx = g0 + 16 * u;
      v[u] = *((ok && gidx < jb.nwg) ? jb.part + ((size_t)gidx * 2 + which) * N + c : zp);
    }
    acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  s[w][lane] = acc;
  __syncthreads();
  if (w == 0 && ok) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s[q][lane];
    float* dst = which ? jb.dbeta : jb.dgamma;
    dst[c] += t;
  }
}

// ---- loss: BCEWithLogits(h)*pen + MSE(v)*pen + MSE(o)*pen, summed over voices, mean over (B,T) ---
// One thread per (row, voice).  stats: [0] loss [1] hit accuracy [3] bce [4] mse_v [5] mse_o.
// d_out = d loss / d (h, v, o)  (w.r.t. the ACTIVATED outputs, like autograd hands them over), or -- with
// WRT_LOGITS -- already multiplied by the head activations' derivative (heads_bwd fused; the train step).
// Reduction across workgroups:
//   TICKET=false: atomicAdd into stats (zeroed by the caller with a memset node);
//   TICKET=true : every workgroup stores its 4 partial sums, the last one to arrive (agent-scope release /
//                 acquire around a ticket counter, cdna_hip_programming.md G16) adds them in a FIXED order ->
//                 bitwise-reproducible stats and no memset node.  The ticket word re-arms itself.
template <bool TICKET, bool WRT_LOGITS>
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ hvo, const float* __restrict__ y, float penalty,
                                                   float* __restrict__ stats, float* __restrict__ d_out, int M,
                                                   float* __restrict__ partials, unsigned* __restrict__ ticket) {
  __shared__ float red[4][4];
  __shared__ int is_last;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const float invM = 1.0f / (float)M;
  float bce = 0.f, mv = 0.f, mo = 0.f, ok = 0.f;
  if (e < M * GT_VOICES) {
    const int m = e / GT_VOICES, j = e % GT_VOICES;
    const size_t base = (size_t)m * GT_TGT + j;
    const float h = hvo[base], v = hvo[base + GT_VOICES], o = hvo[base + 2 * GT_VOICES];
    const float yh = y[base], yv = y[base + GT_VOICES], yo = y[base + 2 * GT_VOICES];
    float gh, gv, go;
    gt_loss_elem<WRT_LOGITS>(h, v, o, yh, yv, yo, penalty, invM, bce, mv, mo, ok, gh, gv, go);
    if (d_out) {
      d_out[base] = gh;
      d_out[base + GT_VOICES] = gv;
      d_out[base + 2 * GT_VOICES] = go;
    }
  }
  bce = gt_wave_sum(bce); mv = gt_wave_sum(mv); mo = gt_wave_sum(mo); ok = gt_wave_sum(ok);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = bce; red[w][1] = mv; red[w][2] = mo; red[w][3] = ok; }
  __syncthreads();
  if (!TICKET) {
    if (threadIdx.x == 0) {
      const float b_ = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) * invM;
      const float v_ = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) * invM;
      const float o_ = (red[0][2] + red[1][2] + red[2][2] + red[3][2]) * invM;
      const float k_ = (red[0][3] + red[1][3] + red[2][3] + red[3][3]) * invM * (1.0f / GT_VOICES);
      atomicAdd(&stats[0], b_ + v_ + o_);
      atomicAdd(&stats[1], k_);
      atomicAdd(&stats[3], b_);
      atomicAdd(&stats[4], v_);
      atomicAdd(&stats[5], o_);
    }
    return;
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) gt_pub_store(partials + blockIdx.x * 4 + q, red[0][q] + red[1][q] + red[2][q] + red[3][q]);
    const unsigned t = gt_pub_ticket(ticket);               // (write-through partials, drained; no fences: gt_common.h)
    is_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  // fixed-order sum: wave q sums quantity q -- lane l takes workgroups l, l+64, ... (all loads in flight at once), then
  // the xor-tree of gt_wave_sum; the order depends only on the grid size, so the stats stay bitwise reproducible
  {
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    float acc = 0.f;
    for (unsigned bk = l; bk < gridDim.x; bk += 64) acc += gt_pub_load(partials + bk * 4 + q);
    acc = gt_wave_sum(acc);
    if (l == 0) red[0][q] = acc * invM;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float b_ = red[0][0], v_ = red[0][1], o_ = red[0][2];
    stats[0] = b_ + v_ + o_;
    stats[1] = red[0][3] * (1.0f / GT_VOICES);
    stats[2] = 0.f;
    stats[3] = b_; stats[4] = v_; stats[5] = o_; stats[6] = 0.f; stats[7] = 0.f;
    *ticket = 0u;                                           // re-arm for the next step
  }
}

// d logits = d (h,v,o) * activation'  (v = sigmoid -> v(1-v);  o = 0.5 tanh -> 0.5 - 2 o^2)
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ d_hvo, const float* __restrict__ hvo,
                                                        float* __restrict__ dlogits, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int c = e % GT_TGT;
  const float a = hvo[e];
  float g = d_hvo[e];
  if (c >= 2 * GT_VOICES) g *= (0.5f - 2.0f * a * a);
  else if (c >= GT_VOICES) g *= a * (1.0f - a);
  dlogits[e] = g;
}

// ---- optimizer: flat multi-tensor update (one launch for all 78+ tensors) -----------------------
// zero_grads: the gradient is consumed and left zeroed (the next backward accumulates into it: no memset node).
// (step / opt_step advance in step_inc_kernel: a last-workgroup ticket inside these kernels was measured SLOWER --
// ~600 arrivals on one counter serialise for longer than the ~4 us a separate launch costs.)
// (gt_bump_counters: gt_common.h)
__global__ void step_inc_kernel(gt_step_state* st, unsigned* err, const float* guard) {
  if (threadIdx.x == 0 && blockIdx.x == 0) gt_bump_counters(st, err, guard);
}

// Fail-safe of the in-launch exchanges (QUAD pair exchange of gt_seq.h, row exchange of gt_gemm64.h): with the region's error word set
// (err; nullptr: this caller has none), or with a non-zero GUARD element g[n - 1] -- padding behind the 27-float output bias, zero in a
// single process; a data-parallel host writes its error flag there before the gradient all-reduce, so every rank sees the sum -- the
// update applies NOTHING: parameters and moments stay, the consumed gradients are cleared.  n - 1 itself is neither updated nor cleared.
// guard = 0 (the public gt_optimizer_step: any n, a sub-range, an unpadded buffer): a plain update of all n elements, no guard, no word.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, int64_t n, const gt_step_state* st,
                                                  int zero_grads, const unsigned* err = nullptr, int guard = 0) {
  const bool skip = (err != nullptr && *err != 0u) || (guard && g[n - 1] != 0.f);
  const float k = skip ? 0.f : st->lr * st->grad_scale;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t nu = guard ? n - 1 : n;                  // elements this update owns
  if (i + 4 <= nu) {
    float4 pv = *reinterpret_cast<float4*>(p + i);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    if (!skip) { pv.x -= k * gv.x; pv.y -= k * gv.y; pv.z -= k * gv.z; pv.w -= k * gv.w; }
    *reinterpret_cast<float4*>(p + i) = pv;
    if (zero_grads) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int64_t j = i; j < nu; ++j) { if (!skip) p[j] -= k * g[j]; if (zero_grads) g[j] = 0.f; }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const gt_step_state* st, int zero_grads,
                                                   int step_advanced, const unsigned* err = nullptr, int guard = 0) {
  const bool skip = (err != nullptr && *err != 0u) || (guard && g[n - 1] != 0.f);        // (see sgd_kernel)
  const int64_t nu = guard ? n - 1 : n;
  const float b1 = st->beta1, b2 = st->beta2, t = (float)(st->opt_step + (step_advanced ? 0u : 1u));
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step_size = st->lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  const float gs = st->grad_scale, eps = st->eps;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = i0 + u;
    if (i < nu) {
      if (!skip) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
      }
      if (zero_grads) g[i] = 0.f;
    }
  }
}

// teacher forcing: tgt_in[b,t] = y[b,t-1], row 0 = zeros
__global__ __launch_bounds__(256) void shift_right_kernel(const float* __restrict__ y, float* __restrict__ tgt, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int m = e / GT_TGT;
  tgt[e] = ((m & 31) == 0) ? 0.f : y[e - GT_TGT];
}

// predict: h = sigmoid(logit) > thres ? 1 : 0 (use_thres 1), the probability (0), or a SAMPLE of it (2: the reference's use_pd --
// h = 1 iff p > u, u = (fmix32((idx * 0x9E3779B1) ^ seed) >> 8) * 2^-24 with idx = the element's flat index m * 27 + c, the hash of
// the dropout masks).  t < 0: all rows (encoder-only); t >= 0: only row t of every sequence, and the step's hits are fed to tgt row
// t+1 (greedy decode).  idx0: flat index of this call's first element inside the whole set (gt_predict_pd_at: a set walked in chunks
// draws the same samples whatever the chunk size).
__global__ __launch_bounds__(256) void predict_head_kernel(const float* __restrict__ hvo_in, float* __restrict__ hvo_out,
                                                           float* __restrict__ tgt, float thres, int use_thres, int t, int B, uint32_t seed,
                                                           uint32_t idx0) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int rows = (t < 0) ? B * 32 : B;
  if (e >= rows * GT_TGT) return;
  const int r = e / GT_TGT, c = e % GT_TGT;
  const int m = (t < 0) ? r : r * 32 + t;
  float a = hvo_in[(size_t)m * GT_TGT + c];
  if (c < GT_VOICES) {
    const float pr = gt_sigmoid(a);
    const float cut = use_thres == 2 ? (float)(gt_fmix32(((idx0 + (uint32_t)(m * GT_TGT + c)) * 0x9E3779B1u) ^ seed) >> 8) * (1.0f / 16777216.0f) : thres;
    a = use_thres ? ((pr > cut) ? 1.0f : 0.0f) : pr;
  }
  hvo_out[(size_t)m * GT_TGT + c] = a;
  if (tgt != nullptr && t >= 0 && t + 1 < 32) tgt[(size_t)(m + 1) * GT_TGT + c] = a;
}

// ---- per-voice evaluation metrics (ref:evaluator.py:522-525 get_hits_accuracies / get_velocity_errors /
// get_micro_timing_errors over the 9 voices of ROLAND_REDUCED_MAPPING) ---------------------------------------------------
// pred / gt: (M,27) HVO = [hits | velocities | offsets].  Column c < 9: 1 if pred == gt (hit agreement), else the squared
// difference.  A workgroup covers 64 rows: thread t < 216 owns column t % 27 of the rows (t / 27) + 8 i -- consecutive threads
// read consecutive floats -- and the 8 row groups are summed in a fixed order, as are the workgroup partials in the second
// kernel: bitwise reproducible.
#define GT_VM_ROWS 64
__global__ __launch_bounds__(256) void voice_metrics_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                   float* __restrict__ part, int M) {
  __shared__ float s[8][GT_TGT];
  const int t = threadIdx.x;
  if (t < 8 * GT_TGT) {
    const int c = t % GT_TGT, g = t / GT_TGT;
    const int m0 = blockIdx.x * GT_VM_ROWS;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < GT_VM_ROWS / 8; ++i) {
      const int m = m0 + g + 8 * i;
      if (m < M) {
        const float a = pred[(size_t)m * GT_TGT + c], b = gt[(size_t)m * GT_TGT + c];
        acc += (c < GT_VOICES) ? ((a == b) ? 1.0f : 0.0f) : (a - b) * (a - b);
      }
    }
    s[g][c] = acc;
  }
  __syncthreads();
  if (t < GT_TGT) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) a += s[g][t];
    part[(size_t)blockIdx.x * GT_TGT + t] = a;
  }
}
// out (30 floats): [0] hit accuracy over all voices, [1..9] per voice; [10] velocity MSE overall, [11..19] per voice;
// [20] offset MSE overall, [21..29] per voice
__global__ __launch_bounds__(64) void voice_metrics_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nwg, int M) {
  __shared__ float s[GT_TGT];
  const int t = threadIdx.x;
  if (t < GT_TGT) {
    float a = 0.f;
    for (int w = 0; w < nwg; ++w) a += part[(size_t)w * GT_TGT + t];
    a /= (float)M;
    s[t] = a;
    out[1 + (t / GT_VOICES) * (GT_VOICES + 1) + t % GT_VOICES] = a;
  }
  __syncthreads();
  if (t < 3) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < GT_VOICES; ++j) a += s[t * GT_VOICES + j];
    out[t * (GT_VOICES + 1)] = a * (1.0f / GT_VOICES);
  }
}

// ---- batch gather (ref:dataset.py:355-356 __getitem__ + the DataLoader's collate, ref:train.py:156-158):
// the dataset stays resident in HBM as two dense tensors; a batch is rows idx[0..B) of each.  One launch fills BOTH static
// step inputs (x (B,32,S) and y (B,32,27)); float4 copies, 16-byte alignment guaranteed by 32 rows per sequence.
__global__ __launch_bounds__(256) void gather_batch_kernel(const float* __restrict__ xs, const float* __restrict__ ys,
                                                           const int64_t* __restrict__ idx, float* __restrict__ x, float* __restrict__ y,
                                                           int B, int S, int64_t n_seq) {
  const int qx = 32 * S / 4, qy = 32 * GT_TGT / 4;             // float4 per sequence
  const int per = qx + qy;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * per) return;
  const int b = (int)(e / per), r = (int)(e % per);
  int64_t src = idx[b];
  src = src < 0 ? 0 : (src >= n_seq ? n_seq - 1 : src);        // a bad index must not read outside the dataset
  if (r < qx) reinterpret_cast<float4*>(x)[(size_t)b * qx + r] = reinterpret_cast<const float4*>(xs)[(size_t)src * qx + r];
  else        reinterpret_cast<float4*>(y)[(size_t)b * qy + (r - qx)] = reinterpret_cast<const float4*>(ys)[(size_t)src * qy + (r - qx)];
}

// ---- bf16 shadows of the encoder layers' weight matrices (gt_config.precision = 1, GemmArgs::B16): for each of the four matrices W
// (R x C) of every layer, W16 = bf16(W) in place order and W16T = bf16(W^T) -- the copy that turns a dgrad (dX = dY W) into the forward's
// NT form.  One workgroup per 32 x 32 tile, transposed through LDS; both outputs leave as 64-byte row segments.  Runs at the head of
// every forward (the optimizer has just rewritten the weights): 2 x 19 MB at d_model 512 / 6 layers, ~10 us.
struct WShadowArgs {
  const float* prm; uint16_t* w16; uint16_t* w16t;
  int64_t in_w, out_w, w1, w2, pstride, sstride;       // layer 0's parameter offsets, floats per layer, bf16 elements per layer of a shadow
  int d, F, L;
};
// The fp32 form (precision = 1 below the shadow threshold): only W^T, in fp32 -- a dgrad then runs as an NT product like the forward.
// At the bf16 MFMA rate the NN form's B fragment (k-strided: eight 4-byte LDS reads per lane) makes a dgrad 1.4-1.5x slower than the
// forward Linear of the same size (C5 bs 64: 18.1 vs 12.2 us per launch).
__global__ __launch_bounds__(256) void weight_transpose_kernel(WShadowArgs a, float* wt) {
  __shared__ float t[32][33];
  const int d = a.d, F = a.F, d32 = d >> 5, f32 = F >> 5;
  const int n0 = 3 * d32 * d32, n1 = d32 * d32, n2 = f32 * d32, T = n0 + n1 + 2 * n2;
  const int l = blockIdx.x / T;
  int f = blockIdx.x % T, R, C;
  int64_t src, so;
  if (f < n0) { src = a.in_w; R = 3 * d; C = d; so = 0; }
  else if (f < n0 + n1) { f -= n0; src = a.out_w; R = d; C = d; so = (int64_t)3 * d * d; }
  else if (f < n0 + n1 + n2) { f -= n0 + n1; src = a.w1; R = F; C = d; so = (int64_t)4 * d * d; }
  else { f -= n0 + n1 + n2; src = a.w2; R = d; C = F; so = (int64_t)4 * d * d + (int64_t)d * F; }
  const int tc = C >> 5, tr = f / tc, tcx = f % tc;
  const float* W = a.prm + src + (int64_t)l * a.pstride;
  float* ot = wt + (int64_t)l * a.sstride + so;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int r = ty + 8 * k; t[r][tx] = W[(size_t)(32 * tr + r) * C + 32 * tcx + tx]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int c = ty + 8 * k; ot[(size_t)(32 * tcx + c) * R + 32 * tr + tx] = t[tx][c]; }
}
__global__ __launch_bounds__(256) void weight_shadow_kernel(WShadowArgs a) {
  __shared__ float t[32][33];
  const int d = a.d, F = a.F, d32 = d >> 5, f32 = F >> 5;
  const int n0 = 3 * d32 * d32, n1 = d32 * d32, n2 = f32 * d32, T = n0 + n1 + 2 * n2;
  const int l = blockIdx.x / T;
  int f = blockIdx.x % T, R, C;
  int64_t src, so;
  if (f < n0) { src = a.in_w; R = 3 * d; C = d; so = 0; }
  else if (f < n0 + n1) { f -= n0; src = a.out_w; R = d; C = d; so = (int64_t)3 * d * d; }
  else if (f < n0 + n1 + n2) { f -= n0 + n1; src = a.w1; R = F; C = d; so = (int64_t)4 * d * d; }
  else { f -= n0 + n1 + n2; src = a.w2; R = d; C = F; so = (int64_t)4 * d * d + (int64_t)d * F; }
  const int tc = C >> 5, tr = f / tc, tcx = f % tc;
  const float* W = a.prm + src + (int64_t)l * a.pstride;
  uint16_t* o = a.w16 + (int64_t)l * a.sstride + so;
  uint16_t* ot = a.w16t + (int64_t)l * a.sstride + so;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = ty + 8 * k;
    const float v = W[(size_t)(32 * tr + r) * C + 32 * tcx + tx];
    t[r][tx] = v;
    o[(size_t)(32 * tr + r) * C + 32 * tcx + tx] = gt_f2bf(v);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = ty + 8 * k;                                  // row of W^T = column of W
    ot[(size_t)(32 * tcx + c) * R + 32 * tr + tx] = gt_f2bf(t[tx][c]);
  }
}

// ---- OutputLayer on a kernel of its own (round 5): hvo[M][27] = act(X[M][d] W^T + b) -- [h logits | sigmoid v | 0.5 tanh o].
// The generic GEMM ran it as 32 x 32 tiles with ONE column tile: M / 32 workgroups (64 at 2048 tokens) each walking d / 64 dependent slabs -- 12-16 us
// for 57 MFLOP at d_model 512.  Here a workgroup owns 16 rows, its four waves split the contraction (d / 4 each), every operand fragment goes
// from global memory straight into registers and ALL of a wave's loads are in flight together (3 d / 64 16-byte loads per lane); the four
// partial 16 x 32 tiles meet in LDS, in a fixed order.  PREC 1: both operands rounded to bf16 (the numbers gt_config.precision >= 1 defines for
// every Linear), products and sums in fp32.  Needs d % 64 == 0, 128 <= d <= 512, M % 16 == 0, dense rows of X (ldx = d).
#define GT_HEADS_MAX_D 512
// L.y != nullptr (the fused train step): the loss of the 16 rows as well -- d loss / d logits to L.dlogits, this workgroup's four partial sums to
// L.partials, the last workgroup to arrive (ticket) adds all partials in a fixed order: loss_kernel<true, true>'s arithmetic and hand-off, one launch less.
struct HeadsLoss { const float* y; float penalty; float* stats; float* partials; unsigned* ticket; float* dlogits; };
template <int PREC>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                        float* __restrict__ hvo, const int M, const int d, const HeadsLoss L) {
  __shared__ __attribute__((aligned(16))) float part[4][16][36];
  __shared__ float outs[16][28];
  __shared__ float red[4][4];
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int kw = d >> 2, k0 = wave * kw;                      // this wave's slice of the contraction
  const float* const zp = gt_zero_ptr();
  constexpr int NKMAX = GT_HEADS_MAX_D / 64;                 // 16-wide k-steps per wave at most
  const int nks = kw >> 4;
  float4 a[NKMAX], b0[NKMAX], b1[NKMAX];
  const float* pa = X + (size_t)(m0 + l16) * d + k0 + 4 * lg;
  const float* pb0 = W + (size_t)l16 * d + k0 + 4 * lg;                                   // columns 0..15
  const float* pb1 = (16 + l16 < GT_TGT) ? W + (size_t)(16 + l16) * d + k0 + 4 * lg : zp;      // columns 16..26 (27..31: zeros)
  const int sb1 = (16 + l16 < GT_TGT) ? 16 : 0;                                           // (the zero page does not advance)
#pragma unroll
  for (int u = 0; u < NKMAX; ++u) {
    if (u < nks) {
      a[u] = *reinterpret_cast<const float4*>(pa + 16 * u);
      b0[u] = *reinterpret_cast<const float4*>(pb0 + 16 * u);
      b1[u] = *reinterpret_cast<const float4*>(pb1 + sb1 * u);
    }
  }
  auto rnd = [](const float v) { return PREC ? gt_bf2f(gt_f2bf(v)) : v; };
  f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
  for (int u = 0; u < NKMAX; ++u) {
    if (u < nks) {
      const float av[4] = {rnd(a[u].x), rnd(a[u].y), rnd(a[u].z), rnd(a[u].w)};
      const float bv0[4] = {rnd(b0[u].x), rnd(b0[u].y), rnd(b0[u].z), rnd(b0[u].w)};
      const float bv1[4] = {rnd(b1[u].x), rnd(b1[u].y), rnd(b1[u].z), rnd(b1[u].w)};
#pragma unroll
      for (int j = 0; j < 4; ++j) { c0 = GT_MFMA16(bv0[j], av[j], c0); c1 = GT_MFMA16(bv1[j], av[j], c1); }
    }
  }
  // lane (l16, lg) holds row l16, columns 4 lg + 0..3 of its column tile (the accumulator map of gt_seq.h's matmul primitives)
  *reinterpret_cast<float4*>(&part[wave][l16][4 * lg]) = make_float4(c0[0], c0[1], c0[2], c0[3]);
  *reinterpret_cast<float4*>(&part[wave][l16][16 + 4 * lg]) = make_float4(c1[0], c1[1], c1[2], c1[3]);
  __syncthreads();
  for (int e = tid; e < 16 * GT_TGT; e += 256) {
    const int r = e / GT_TGT, c = e % GT_TGT;
    float v = ((part[0][r][c] + part[1][r][c]) + (part[2][r][c] + part[3][r][c])) + bias[c];
    if (c >= 2 * GT_VOICES) v = 0.5f * tanhf(v);
    else if (c >= GT_VOICES) v = gt_sigmoid(v);
    hvo[(size_t)(m0 + r) * GT_TGT + c] = v;
    outs[r][c] = v;
  }
  if (L.y == nullptr) return;                                // (uniform)
  __syncthreads();
  const float invM = 1.0f / (float)M;
  float bce = 0.f, mv = 0.f, mo = 0.f, ok = 0.f;
  if (tid < 16 * GT_VOICES) {
    const int r = tid / GT_VOICES, j = tid % GT_VOICES;
    const size_t base = (size_t)(m0 + r) * GT_TGT + j;
    float gh, gv, go;
    gt_loss_elem<true>(outs[r][j], outs[r][j + GT_VOICES], outs[r][j + 2 * GT_VOICES], L.y[base], L.y[base + GT_VOICES], L.y[base + 2 * GT_VOICES],
                       L.penalty, invM, bce, mv, mo, ok, gh, gv, go);
    L.dlogits[base] = gh; L.dlogits[base + GT_VOICES] = gv; L.dlogits[base + 2 * GT_VOICES] = go;
  }
  bce = gt_wave_sum(bce); mv = gt_wave_sum(mv); mo = gt_wave_sum(mo); ok = gt_wave_sum(ok);
  if (lane == 0) { red[wave][0] = bce; red[wave][1] = mv; red[wave][2] = mo; red[wave][3] = ok; }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) gt_pub_store(L.partials + blockIdx.x * 4 + q, (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]));
    const unsigned t = gt_pub_ticket(L.ticket);               // (write-through partials, drained; no fences: gt_common.h)
    is_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  {                                                          // fixed-order sum: wave q sums quantity q over the workgroups (loss_kernel)
    float acc = 0.f;
    for (unsigned bk = lane; bk < gridDim.x; bk += 64) acc += gt_pub_load(L.partials + bk * 4 + wave);
    acc = gt_wave_sum(acc);
    if (lane == 0) red[0][wave] = acc * invM;
  }
  __syncthreads();
  if (tid == 0) {
    const float b_ = red[0][0], v_ = red[0][1], o_ = red[0][2];
    L.stats[0] = b_ + v_ + o_;
    L.stats[1] = red[0][3] * (1.0f / GT_VOICES);
    L.stats[2] = 0.f;
    L.stats[3] = b_; L.stats[4] = v_; L.stats[5] = o_; L.stats[6] = 0.f; L.stats[7] = 0.f;
    *L.ticket = 0u;                                          // re-arm for the next step
  }
}
static inline bool heads_fwd_ok(int M, int d, int ldx, const void* X, const void* W) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return M % 16 == 0 && d % 64 == 0 && d >= 128 && d <= GT_HEADS_MAX_D && ldx == d && al16(X) && al16(W);
}
