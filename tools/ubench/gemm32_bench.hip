// Prototype bench: fp32 NT GEMM C[M,N] = A[M,K] B[N,K]^T + bias on v_mfma_f32_32x32x2_f32, 128x128 tiles, and the in-kernel
// clock the chip holds while it runs (s_memtime / s_memrealtime stamps into a buffer nothing else reads).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/gemm32_bench.hip -o tools/ubench/gemm32_bench
//   tools/ubench/gemm32_bench [M N K]        (default: the d_model 512 QKV projection at 16384 tokens)
// Why 32x32x2: lane (r = l & 31, h = l >> 5) reads one float4 (k = 4h .. 4h+3) per 32-row fragment, and with that lane->row map
// a row stride of BK + 4 floats is conflict-free for ds_read_b128 (the 16x16x4 map needs BK + 8): 72 KB of LDS per 128x128x32
// double-buffered workgroup instead of 80 KB, so TWO workgroups fit a CU with room to spare and one's prologue / barriers /
// epilogue hide behind the other's MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BK, int WAVES_PER_SIMD, bool STAMP>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void gemm32_nt(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                                  const float* __restrict__ bias, int M, int N, int K, long long* stamps) {
  constexpr int BM = 128, BN = 128, STR = BK + 4, TSZ = BM * STR;
  __shared__ __attribute__((aligned(16))) float smem[4 * TSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  // XCD-contiguous tile order (placement only changes speed)
  const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const int xcd = lin & 7, q = nb >> 3, rr = nb & 7;
  const int b = xcd * q + (xcd < rr ? xcd : rr) + (lin >> 3);
  const int m0 = (b / gx) * BM, n0 = (b % gx) * BN;
  long long t0 = 0, w0 = 0;
  if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); w0 = __builtin_amdgcn_s_memrealtime(); }

  constexpr int CPR = BK / 4, PER = BM * CPR / 256;     // float4 chunks per row / per thread per operand
  f32x4 va[PER], vb[PER];
  const char* pa[PER];
  const char* pb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;
    pa[i] = reinterpret_cast<const char*>(A + (size_t)(m0 + r) * K + c);
    pb[i] = reinterpret_cast<const char*>(B + (size_t)(n0 + r) * K + c);
  }
#define LOAD_SLAB(k0)                                                                          \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    va[i] = *reinterpret_cast<const f32x4*>(pa[i] + (size_t)(k0) * 4);                         \
    vb[i] = *reinterpret_cast<const f32x4*>(pb[i] + (size_t)(k0) * 4);                         \
  }
#define STORE_SLAB(buf)                                                                        \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;                            \
    *reinterpret_cast<f32x4*>(&smem[(buf) * TSZ + r * STR + c]) = va[i];                       \
    *reinterpret_cast<f32x4*>(&smem[2 * TSZ + (buf) * TSZ + r * STR + c]) = vb[i];             \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BK;
  LOAD_SLAB(0)
  STORE_SLAB(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const float* sa = smem + cur * TSZ + (wm * 64 + r32) * STR + 4 * h;
    const float* sb = smem + 2 * TSZ + cur * TSZ + (wn * 64 + r32) * STR + 4 * h;
    if (kt + 1 < nk) { LOAD_SLAB((kt + 1) * BK) }
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      if (kk == BK / 16 && kt + 1 < nk) { STORE_SLAB(cur ^ 1) }     // mid-slab hand-over: the closing barrier waits on nothing
      f32x4 af[2], bf[2];
      af[0] = *reinterpret_cast<const f32x4*>(sa + kk * 8); af[1] = *reinterpret_cast<const f32x4*>(sa + 32 * STR + kk * 8);
      bf[0] = *reinterpret_cast<const f32x4*>(sb + kk * 8); bf[1] = *reinterpret_cast<const f32x4*>(sb + 32 * STR + kk * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)     // transposed tile: lane holds ONE row of C and 4-column groups -> 16-byte stores
            acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[tb][j], af[ta][j], acc[ta][tb], 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: lane (r32, h) holds row m0 + wm 64 + ta 32 + r32, columns n0 + wn 64 + tb 32 + 8 g + 4 h + 0..3 in registers 4g..4g+3
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = n0 + wn * 64 + tb * 32 + 8 * g + 4 * h;
      const float4 bi = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        const int row = m0 + wm * 64 + ta * 32 + r32;
        *reinterpret_cast<float4*>(&C[(size_t)row * N + col]) =
            make_float4(acc[ta][tb][4 * g] + bi.x, acc[ta][tb][4 * g + 1] + bi.y, acc[ta][tb][4 * g + 2] + bi.z, acc[ta][tb][4 * g + 3] + bi.w);
      }
    }
  if (STAMP && tid == 0) {
    stamps[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = __builtin_amdgcn_s_memtime() - t0;
    stamps[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime() - w0;
  }
}

// Persistent form: 2 workgroups per CU walk the tile list; the first slab of the NEXT tile is requested before the epilogue of
// the current one, so its latency hides behind the stores (tile prologues are ~4 % of a K = 512 tile).
template <int BK, int MODE>
__global__ __launch_bounds__(256, 2) void gemm32_nt_persist(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                            const float* __restrict__ bias, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, STR = BK + 4, TSZ = BM * STR;
  __shared__ __attribute__((aligned(16))) float smem[4 * TSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int gx = N / BN, ntiles = gx * (M / BM);
  constexpr int CPR = BK / 4, PER = BM * CPR / 256;
  f32x4 va[PER], vb[PER];
  uint32_t roff[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;
    roff[i] = (uint32_t)(r * K + c) * 4u;
  }
  const int nk = K / BK;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // de-phase the two workgroups that share a CU (they run the same program at the same speed: in lockstep both stall at
  // the same moment and the matrix pipe idles).  MODE 1/3: static priority for one of them; MODE 2: a one-off delay.
  if (MODE == 1 && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(1);
  if (MODE == 3 && (blockIdx.x & 1)) __builtin_amdgcn_s_setprio(1);
  if (MODE == 2 && blockIdx.x >= gridDim.x / 2) { __builtin_amdgcn_s_sleep(64); __builtin_amdgcn_s_sleep(64); }
  const char* ta_ = reinterpret_cast<const char*>(A + (size_t)(t / gx) * BM * K);
  const char* tb_ = reinterpret_cast<const char*>(B + (size_t)(t % gx) * BN * K);
#define PLOAD(k0)                                                                              \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    va[i] = *reinterpret_cast<const f32x4*>(ta_ + roff[i] + (size_t)(k0) * 4);                 \
    vb[i] = *reinterpret_cast<const f32x4*>(tb_ + roff[i] + (size_t)(k0) * 4);                 \
  }
  PLOAD(0)
  STORE_SLAB(0)
  __syncthreads();
  for (;;) {
    const int m0 = (t / gx) * BM, n0 = (t % gx) * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int tn = t + gridDim.x;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const float* sa = smem + cur * TSZ + (wm * 64 + r32) * STR + 4 * h;
      const float* sb = smem + 2 * TSZ + cur * TSZ + (wn * 64 + r32) * STR + 4 * h;
      if (kt + 1 < nk) { PLOAD((kt + 1) * BK) }
      else if (tn < ntiles) {                       // last slab: request the next tile's first slab
        ta_ = reinterpret_cast<const char*>(A + (size_t)(tn / gx) * BM * K);
        tb_ = reinterpret_cast<const char*>(B + (size_t)(tn % gx) * BN * K);
        PLOAD(0)
      }
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        if (kk == BK / 16 && kt + 1 < nk) { STORE_SLAB(cur ^ 1) }
        f32x4 af[2], bf[2];
        af[0] = *reinterpret_cast<const f32x4*>(sa + kk * 8); af[1] = *reinterpret_cast<const f32x4*>(sa + 32 * STR + kk * 8);
        bf[0] = *reinterpret_cast<const f32x4*>(sb + kk * 8); bf[1] = *reinterpret_cast<const f32x4*>(sb + 32 * STR + kk * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
              acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[tb][j], af[ta][j], acc[ta][tb], 0, 0, 0);
      }
      __syncthreads();
    }
    // (nk is even for K = 512 / BK = 32: the last slab sat in buffer 1, buffer 0 is free for the next tile's first slab)
    if (tn < ntiles) { STORE_SLAB(0) }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = n0 + wn * 64 + tb * 32 + 8 * g + 4 * h;
        const float4 bi = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
          const int row = m0 + wm * 64 + ta * 32 + r32;
          *reinterpret_cast<float4*>(&C[(size_t)row * N + col]) =
              make_float4(acc[ta][tb][4 * g] + bi.x, acc[ta][tb][4 * g + 1] + bi.y, acc[ta][tb][4 * g + 2] + bi.z, acc[ta][tb][4 * g + 3] + bi.w);
        }
      }
    if (tn >= ntiles) break;
    t = tn;
    __syncthreads();
  }
}

// Software-pipelined fragment reads: the LDS reads of k-step kk+1 are issued BEFORE the 16 MFMAs of k-step kk, and the slab
// hand-over (store of slab t+1, barrier) sits between k-steps 2 and 3 so that the first fragments of slab t+1 are read under
// the last 16 MFMAs of slab t.  The MFMA stream of a wave then never waits for LDS (the plain loop exposes one LDS round
// trip per fragment group: ~15 % of a slab).
// ABL (ablation, results then wrong by construction): 1 = no global loads / LDS stores inside the loop (slab 0 reused),
// 2 = also no LDS fragment reads inside the loop, 3 = as 1 and no barrier, 4 = no epilogue stores
template <int SCHED, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm32_nt_pipe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         const float* __restrict__ bias, int M, int N, int K) {
  constexpr int BK = 32, BM = 128, BN = 128, STR = BK + 4, TSZ = BM * STR;
  __shared__ __attribute__((aligned(16))) float smem[4 * TSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const int xcd = lin & 7, q = nb >> 3, rr = nb & 7;
  const int b = xcd * q + (xcd < rr ? xcd : rr) + (lin >> 3);
  const int m0 = (b / gx) * BM, n0 = (b % gx) * BN;
  constexpr int CPR = BK / 4, PER = BM * CPR / 256;
  f32x4 va[PER], vb[PER];
  const char* pa[PER];
  const char* pb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;
    pa[i] = reinterpret_cast<const char*>(A + (size_t)(m0 + r) * K + c);
    pb[i] = reinterpret_cast<const char*>(B + (size_t)(n0 + r) * K + c);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
  const int offa = (wm * 64 + r32) * STR + 4 * h, offb = 2 * TSZ + (wn * 64 + r32) * STR + 4 * h;
#define RD(FA, FB, buf, kk)                                                                   \
  FA[0] = *reinterpret_cast<const f32x4*>(smem + (buf) * TSZ + offa + (kk) * 8);              \
  FA[1] = *reinterpret_cast<const f32x4*>(smem + (buf) * TSZ + offa + 32 * STR + (kk) * 8);   \
  FB[0] = *reinterpret_cast<const f32x4*>(smem + (buf) * TSZ + offb + (kk) * 8);              \
  FB[1] = *reinterpret_cast<const f32x4*>(smem + (buf) * TSZ + offb + 32 * STR + (kk) * 8);
#define MM(FA, FB)                                                                            \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                               \
  _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                            \
  _Pragma("unroll") for (int tb = 0; tb < 2; ++tb)                                            \
    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(FB[tb][j], FA[ta][j], acc[ta][tb], 0, 0, 0);
#define SB() if (SCHED) __builtin_amdgcn_sched_barrier(0);
  const int nk = K / BK;
  LOAD_SLAB(0)
  STORE_SLAB(0)
  __syncthreads();
  RD(fa0, fb0, 0, 0)
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    const bool mem = more && (ABL == 0 || ABL >= 4);
    if (mem) { if (ABL == 5) { LOAD_SLAB(0) } else { LOAD_SLAB((kt + 1) * BK) } }
    SB()
    if (ABL != 2) { RD(fa1, fb1, cur, 1) }
    MM(fa0, fb0)
    SB()
    if (ABL != 2) { RD(fa0, fb0, cur, 2) }
    MM(fa1, fb1)
    SB()
    if (mem && ABL != 6) { STORE_SLAB(cur ^ 1) }
    if (mem && ABL == 6) {
#pragma unroll
      for (int i = 0; i < PER; ++i) asm volatile("" :: "v"(va[i]), "v"(vb[i]));
    }
    if (ABL != 2) { RD(fa1, fb1, cur, 3) }
    MM(fa0, fb0)
    SB()
    if (ABL != 3) __syncthreads();
    if (more && ABL != 2) { RD(fa0, fb0, (ABL == 0 || ABL == 4 || ABL == 5) ? (cur ^ 1) : 0, 0) }
    SB()
    MM(fa1, fb1)
    SB()
  }
  if (ABL == 2) { RD(fa1, fb1, 0, 1) }
  if (ABL == 4) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 123.456f) C[0] = t;
    return;
  }
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = n0 + wn * 64 + tb * 32 + 8 * g + 4 * h;
      const float4 bi = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        const int row = m0 + wm * 64 + ta * 32 + r32;
        *reinterpret_cast<float4*>(&C[(size_t)row * N + col]) =
            make_float4(acc[ta][tb][4 * g] + bi.x, acc[ta][tb][4 * g + 1] + bi.y, acc[ta][tb][4 * g + 2] + bi.z, acc[ta][tb][4 * g + 3] + bi.w);
      }
    }
}

// Two-deep register ring: the global loads of slab t+2 are issued at the top of slab t, slab t+1's registers go to LDS in the
// middle of slab t -- every load has 1.5 slab times (~2.6 us) to come back and ~64 KB per workgroup pair stay in flight all the
// time (Little: 37 GB/s per CU at 1-1.5 us loaded latency needs ~55 KB in flight; the one-deep form has ~32 KB on average).
template <int WPS>
__global__ __launch_bounds__(256, WPS) void gemm32_nt_ring(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                          const float* __restrict__ bias, int M, int N, int K) {
  constexpr int BK = 32, BM = 128, BN = 128, STR = BK + 4, TSZ = BM * STR;
  __shared__ __attribute__((aligned(16))) float smem[4 * TSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const int xcd = lin & 7, q = nb >> 3, rr = nb & 7;
  const int b = xcd * q + (xcd < rr ? xcd : rr) + (lin >> 3);
  const int m0 = (b / gx) * BM, n0 = (b % gx) * BN;
  constexpr int CPR = BK / 4, PER = BM * CPR / 256;
  f32x4 va[PER], vb[PER], wa[PER], wb[PER];
  const char* pa[PER];
  const char* pb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;
    pa[i] = reinterpret_cast<const char*>(A + (size_t)(m0 + r) * K + c);
    pb[i] = reinterpret_cast<const char*>(B + (size_t)(n0 + r) * K + c);
  }
#define LD2(XA, XB, k0)                                                                        \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    XA[i] = *reinterpret_cast<const f32x4*>(pa[i] + (size_t)(k0) * 4);                         \
    XB[i] = *reinterpret_cast<const f32x4*>(pb[i] + (size_t)(k0) * 4);                         \
  }
#define ST2(XA, XB, buf)                                                                       \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;                            \
    *reinterpret_cast<f32x4*>(&smem[(buf) * TSZ + r * STR + c]) = XA[i];                       \
    *reinterpret_cast<f32x4*>(&smem[2 * TSZ + (buf) * TSZ + r * STR + c]) = XB[i];             \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
  const int offa = (wm * 64 + r32) * STR + 4 * h, offb = 2 * TSZ + (wn * 64 + r32) * STR + 4 * h;
  const int nk = K / BK;                       // even, >= 2
  LD2(va, vb, 0)
  LD2(wa, wb, BK)                               // slab 1 -> set w
  ST2(va, vb, 0)
  __syncthreads();
  RD(fa0, fb0, 0, 0)
  // one slab: CUR = LDS buffer of slab t; (NA, NB) = the register set holding slab t+1; (FA, FB) = the set slab t used (free):
  // it receives slab t+2
#define SLAB(CUR, NA, NB, FA_, FB_, t)                                                         \
  if ((t) + 2 < nk) { LD2(FA_, FB_, ((t) + 2) * BK) }                                          \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  RD(fa1, fb1, CUR, 1) MM(fa0, fb0)                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  RD(fa0, fb0, CUR, 2) MM(fa1, fb1)                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  if ((t) + 1 < nk) { ST2(NA, NB, (CUR) ^ 1) }                                                 \
  RD(fa1, fb1, CUR, 3) MM(fa0, fb0)                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  __syncthreads();                                                                             \
  if ((t) + 1 < nk) { RD(fa0, fb0, (CUR) ^ 1, 0) }                                             \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  MM(fa1, fb1)                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; kt += 2) {
    SLAB(0, wa, wb, va, vb, kt)
    SLAB(1, va, vb, wa, wb, kt + 1)
  }
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = n0 + wn * 64 + tb * 32 + 8 * g + 4 * h;
      const float4 bi = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        const int row = m0 + wm * 64 + ta * 32 + r32;
        *reinterpret_cast<float4*>(&C[(size_t)row * N + col]) =
            make_float4(acc[ta][tb][4 * g] + bi.x, acc[ta][tb][4 * g + 1] + bi.y, acc[ta][tb][4 * g + 2] + bi.z, acc[ta][tb][4 * g + 3] + bi.w);
      }
    }
}

// Two-deep register ring: the global loads of slab t+2 are issued at the top of slab t, slab t+1's registers go to LDS in the
// middle of slab t -- every load has 1.5 slab times (~2.6 us) to come back and ~64 KB per workgroup pair stay in flight all the
// time (Little: 37 GB/s per CU at 1-1.5 us loaded latency needs ~55 KB in flight; the one-deep form has ~32 KB on average).
// as above + branch-free slab body and compile-time interleave (sched_group_barrier): one VMEM / DS instruction behind each
// MFMA instead of blocks of 8 that hold the wave's issue port while the matrix pipe runs dry
template <int WPS>
__global__ __launch_bounds__(256, WPS) void gemm32_nt_ring2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                          const float* __restrict__ bias, int M, int N, int K) {
  constexpr int BK = 32, BM = 128, BN = 128, STR = BK + 4, TSZ = BM * STR;
  __shared__ __attribute__((aligned(16))) float smem[4 * TSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const int xcd = lin & 7, q = nb >> 3, rr = nb & 7;
  const int b = xcd * q + (xcd < rr ? xcd : rr) + (lin >> 3);
  const int m0 = (b / gx) * BM, n0 = (b % gx) * BN;
  constexpr int CPR = BK / 4, PER = BM * CPR / 256;
  f32x4 va[PER], vb[PER], wa[PER], wb[PER];
  const char* pa[PER];
  const char* pb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;
    pa[i] = reinterpret_cast<const char*>(A + (size_t)(m0 + r) * K + c);
    pb[i] = reinterpret_cast<const char*>(B + (size_t)(n0 + r) * K + c);
  }
#define LD2(XA, XB, k0)                                                                        \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    XA[i] = *reinterpret_cast<const f32x4*>(pa[i] + (size_t)(k0) * 4);                         \
    XB[i] = *reinterpret_cast<const f32x4*>(pb[i] + (size_t)(k0) * 4);                         \
  }
#define ST2(XA, XB, buf)                                                                       \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    const int ch = tid + i * 256, r = ch / CPR, c = (ch % CPR) * 4;                            \
    *reinterpret_cast<f32x4*>(&smem[(buf) * TSZ + r * STR + c]) = XA[i];                       \
    *reinterpret_cast<f32x4*>(&smem[2 * TSZ + (buf) * TSZ + r * STR + c]) = XB[i];             \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
  const int offa = (wm * 64 + r32) * STR + 4 * h, offb = 2 * TSZ + (wn * 64 + r32) * STR + 4 * h;
  const int nk = K / BK;                       // even, >= 2
  LD2(va, vb, 0)
  LD2(wa, wb, BK)                               // slab 1 -> set w
  ST2(va, vb, 0)
  __syncthreads();
  RD(fa0, fb0, 0, 0)
  // one slab: CUR = LDS buffer of slab t; (NA, NB) = the register set holding slab t+1; (FA, FB) = the set slab t used (free):
  // it receives slab t+2
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define IL(mask, n) _Pragma("unroll") for (int z_ = 0; z_ < (n); ++z_) { SGB(0x8, 1) SGB(mask, 1) }
#define SLAB2(CUR, NA, NB, FA_, FB_, t)                                                        \
  { const int k2_ = ((t) + 2 < nk ? (t) + 2 : nk - 1) * BK;                                    \
    LD2(FA_, FB_, k2_) }                                                                       \
  RD(fa1, fb1, CUR, 1) MM(fa0, fb0)                                                            \
  IL(0x20, 8) IL(0x100, 4) SGB(0x8, 4)                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  RD(fa0, fb0, CUR, 2) MM(fa1, fb1)                                                            \
  IL(0x100, 4) SGB(0x8, 12)                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  ST2(NA, NB, (CUR) ^ 1)                                                                       \
  RD(fa1, fb1, CUR, 3) MM(fa0, fb0)                                                            \
  IL(0x200, 8) IL(0x100, 4) SGB(0x8, 4)                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  __syncthreads();                                                                             \
  RD(fa0, fb0, (CUR) ^ 1, 0)                                                                   \
  MM(fa1, fb1)                                                                                 \
  IL(0x100, 4) SGB(0x8, 12)                                                                    \
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; kt += 2) {
    SLAB2(0, wa, wb, va, vb, kt)
    SLAB2(1, va, vb, wa, wb, kt + 1)
  }
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = n0 + wn * 64 + tb * 32 + 8 * g + 4 * h;
      const float4 bi = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        const int row = m0 + wm * 64 + ta * 32 + r32;
        *reinterpret_cast<float4*>(&C[(size_t)row * N + col]) =
            make_float4(acc[ta][tb][4 * g] + bi.x, acc[ta][tb][4 * g + 1] + bi.y, acc[ta][tb][4 * g + 2] + bi.z, acc[ta][tb][4 * g + 3] + bi.w);
      }
    }
}

template <int BK, int W, bool STAMP>
static double run(const char* tag, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long long* stamps, int reps) {
  dim3 grid(N / 128, M / 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) gemm32_nt<BK, W, STAMP><<<grid, 256>>>(A, B, C, bias, M, N, K, stamps);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) gemm32_nt<BK, W, STAMP><<<grid, 256>>>(A, B, C, bias, M, N, K, stamps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, tf = 2.0 * M * N * K / us / 1e6;
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm32_nt<BK, W, STAMP>, 256, 0);
  printf("%-40s grid %4d x %3d  %8.1f us  %6.1f TFLOP/s (%.1f %% of 157.3)  resident WG/CU %d", tag, grid.y, grid.x, us, tf, 100.0 * tf / 157.3, occ);
  if (STAMP) {
    std::vector<long long> h(2 * grid.x * grid.y);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (size_t i = 0; i < h.size() / 2; ++i) if (h[2 * i + 1] > 0) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);   // GHz: memrealtime ticks at 100 MHz
    std::sort(clk.begin(), clk.end());
    printf("  in-kernel clock median %.2f GHz", clk.empty() ? 0.0 : clk[clk.size() / 2]);
  }
  printf("\n");
  return tf;
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 512;
  float *A, *B, *C, *bias; long long* stamps;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&bias, (size_t)N * 4);
  hipMalloc(&stamps, (size_t)(M / 128) * (N / 128) * 16);
  std::vector<float> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.0f - 0.5f;
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data() + 12345, (size_t)N * K * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, (size_t)N * 4);
  printf("C[%d,%d] = A[%d,%d] * B[%d,%d]^T  fp32, v_mfma_f32_32x32x2_f32, 128x128 tiles\n", M, N, M, K, N, K);
  // correctness spot check against a host dot product
  gemm32_nt<32, 2, false><<<dim3(N / 128, M / 128), 256>>>(A, B, C, bias, M, N, K, stamps);
  std::vector<float> c(1536);
  const int rows[4] = {0, 77, 4099, M - 1};
  double worst = 0;
  for (int ri = 0; ri < 4; ++ri) {
    hipMemcpy(c.data(), C + (size_t)rows[ri] * N, std::min(N, 1536) * 4, hipMemcpyDeviceToHost);
    for (int n = 0; n < std::min(N, 1536); n += 37) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)h[(size_t)rows[ri] * K + k] * (double)h[12345 + (size_t)n * K + k];
      worst = std::max(worst, fabs(s - c[n]));
    }
  }
  printf("spot check vs host fp64: max |diff| %.3g %s\n", worst, worst < 1e-4 ? "ok" : "WRONG");
  const int reps = 20;
  run<32, 2, false>("BK32, 2 waves/SIMD (2 WG/CU)", A, B, C, bias, M, N, K, stamps, reps);
  run<32, 1, false>("BK32, 1 wave/SIMD bound", A, B, C, bias, M, N, K, stamps, reps);
  run<16, 2, false>("BK16, 2 waves/SIMD", A, B, C, bias, M, N, K, stamps, reps);
  run<64, 1, false>("BK64, 1 wave/SIMD (147 KB LDS)", A, B, C, bias, M, N, K, stamps, reps);
  run<32, 2, true>("BK32, 2 waves/SIMD, clock stamps", A, B, C, bias, M, N, K, stamps, reps);
  if ((K / 32) % 2 == 0) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto bench = [&](auto kern, const char* tag, int ng) {
      hipMemset(C, 0, (size_t)M * N * 4);
      for (int i = 0; i < 3; ++i) kern<<<ng, 256>>>(A, B, C, bias, M, N, K);
      hipEventRecord(a);
      for (int i = 0; i < reps; ++i) kern<<<ng, 256>>>(A, B, C, bias, M, N, K);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double us = 1e3 * ms / reps, tf = 2.0 * M * N * K / us / 1e6;
      float c0 = 0; hipMemcpy(&c0, C + (size_t)(M - 1) * N + N - 1, 4, hipMemcpyDeviceToHost);
      printf("persistent BK32 %-22s %3d WGs  %8.1f us  %6.1f TFLOP/s (%.1f %% of 157.3)   C[last]=%g\n", tag, ng, us, tf, 100.0 * tf / 157.3, c0);
    };
    auto bench2 = [&](auto kern, const char* tag) {
      hipMemset(C, 0, (size_t)M * N * 4);
      dim3 grid(N / 128, M / 128);
      for (int i = 0; i < 3; ++i) kern<<<grid, 256>>>(A, B, C, bias, M, N, K);
      hipEventRecord(a);
      for (int i = 0; i < reps; ++i) kern<<<grid, 256>>>(A, B, C, bias, M, N, K);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double us = 1e3 * ms / reps, tf = 2.0 * M * N * K / us / 1e6;
      float c0 = 0; hipMemcpy(&c0, C + (size_t)(M - 1) * N + N - 1, 4, hipMemcpyDeviceToHost);
      printf("%-44s  %8.1f us  %6.1f TFLOP/s (%.1f %% of 157.3)   C[last]=%g\n", tag, us, tf, 100.0 * tf / 157.3, c0);
    };
    for (int round = 0; round < 2; ++round) {
      bench2(gemm32_nt_pipe<0>, "pipelined fragment reads (compiler order)");
      bench2(gemm32_nt_pipe<1>, "pipelined fragment reads (pinned phases)");
      bench2(gemm32_nt_ring<2>, "two-deep register ring, 2 waves/SIMD");
      bench2(gemm32_nt_ring<1>, "two-deep register ring, 1 wave/SIMD bound");
      bench2(gemm32_nt_ring2<2>, "ring + branch-free + interleave, 2 waves/SIMD");
      bench2(gemm32_nt_ring2<1>, "ring + branch-free + interleave, 1 wave/SIMD bound");
      bench2(gemm32_nt_pipe<1, 1>, "  ablation: no global loads / LDS stores in loop");
      bench2(gemm32_nt_pipe<1, 2>, "  ablation: + no LDS fragment reads");
      bench2(gemm32_nt_pipe<1, 3>, "  ablation: no loads, no barrier");
      bench2(gemm32_nt_pipe<1, 4>, "  ablation: no epilogue stores");
      bench2(gemm32_nt_pipe<1, 5>, "  ablation: loads always from slab 0 (cache hits)");
      bench2(gemm32_nt_pipe<1, 6>, "  ablation: loads kept, no LDS stores");
      bench(gemm32_nt_persist<32, 0>, "plain", 256);
      bench(gemm32_nt_persist<32, 0>, "plain", 512);
      bench(gemm32_nt_persist<32, 1>, "prio upper half", 512);
      bench(gemm32_nt_persist<32, 3>, "prio odd", 512);
      bench(gemm32_nt_persist<32, 2>, "stagger upper half", 512);
    }
  }
  return 0;
}
