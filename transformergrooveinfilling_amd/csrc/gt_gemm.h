// fp32 MFMA tile GEMM with fused epilogues -- the dominant kernel family of the train step.
//
//   C[M,N] (+)= sum_k A(m,k) * B(k,n)
//
// Operand storage (template flags):
//   AKM=false : A(m,k) = A[m*lda + k]   (k contiguous; activations / incoming gradients)
//   AKM=true  : A(m,k) = A[k*lda + m]   (m contiguous; wgrad: A = dY^T)
//   BKM=false : B(k,n) = B[n*ldb + k]   (torch Linear weight (out,in) used as x W^T: forward "NT")
//   BKM=true  : B(k,n) = B[k*ldb + n]   (dgrad "NN": dX = dY W;  wgrad "TN": dW = dY^T X)
//
// Workgroup = WM x WN waves; each wave owns TM x TN tiles of v_mfma_f32_16x16x4_f32 (exact fp32;
// 64 FLOP/clk/SIMD = the chip's 157.3 TF fp32 matrix peak, MI355X_MICROARCH.md "Matrix cores").
// K is walked in BK-wide slabs (64 by default: few, long slabs keep the latency-exposed
// load->barrier hand-offs rare on these short-K GEMMs), double-buffered in LDS with register staging (global loads of slab
// t+1 are in flight while slab t is multiplied; one barrier per slab).  Tiles are kept in LDS in
// SOURCE orientation so both global->LDS copies are 16-byte vector moves:
//   k-contiguous source : s[row][BK+8]  -> fragment for 4 MFMA k-steps = ONE ds_read_b128
//                         (lane (r=l&15, g=l>>4) reads k = 4g..4g+3; A and B use the same
//                         k permutation, so the products pair up correctly)
//   row-contiguous source: s[k][rows+4] -> fragment = 4 ds_read_b32 (stride%8==4: conflict-free)
// Out-of-range rows/cols/k are zero-filled on load and masked on store, so every M, N, K works
// (S=27 inputs, 27-wide HVO heads, d_model=16...).
//
// Row epilogues (EPI_RES_LN / EPI_RES_LNBWD) need BN >= N: the workgroup owns whole rows, stages
// the accumulators through LDS and runs LayerNorm forward/backward with 64-lane wave reductions.
#pragma once
#include "gt_common.h"
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

#ifndef GT_MIDSLAB_STORE
#define GT_MIDSLAB_STORE 1
#endif
enum {
  EPI_STORE = 0,          // C = acc + bias (+ C if accumulate)
  EPI_ATOMIC = 1,         // atomicAdd(C, acc)            (split-K wgrad; + bias-grad column sums of A)
  EPI_RELU_PE = 2,        // aux = acc+bias; C = drop(relu(aux) + pe[t])          (InputLayer)
  EPI_RELU_DROP = 3,      // C = drop(relu(acc+bias))                               (FFN linear1)
  EPI_HEADS = 4,          // C = [h | sigmoid v | 0.5 tanh o](acc+bias)             (OutputLayer)
  EPI_MASK_NZ = 5,        // C = acc * (res != 0 ? mask_scale : 0)                  (FFN linear2 dgrad)
  EPI_ADD_RELUMASK_DROP = 6,  // C = (acc + res) * dropmask * (aux_in > 0)          (InputLayer backward)
  EPI_RES_LN = 7,         // z = drop(acc+bias) + res; C = LN(z); aux = xhat; aux2 = rstd
  EPI_RES_LNBWD = 8,      // g = acc (+ res); C = LNbwd(g); C2 = C*dropmask; dgamma/dbeta atomics
  EPI_RES_LN_X = 9,       // EPI_RES_LN / EPI_RES_LNBWD on 32x32 tiles that do NOT own their rows: the row partials of the N / 32 column tiles
  EPI_RES_LNBWD_X = 10    // meet through the in-launch row exchange (gt_gemm64.h, gemm_xln32_epilogue; round 6)
};

struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K, lda, ldb, ldc;
  int k_chunk;               // K range per blockIdx.z (multiple of 16)
  int accumulate;
  const float* bias;         // [N] or nullptr
  const float* res; int ldres;
  const float* aux_in;       // a0 for EPI_ADD_RELUMASK_DROP (ld = N)
  float* aux;                // [M,N] (ld = N): a0 / xhat
  float* aux2;               // [M]: rstd
  float* C2;                 // second output (ld = ldc) or nullptr
  const float* gamma; const float* beta;
  const float* xhat; const float* rstd;     // LN backward inputs (ld = N)
  float* dgamma; float* dbeta;
  float* ln_part;            // if set: per-row-tile partials [row_tile][2][N] instead of contended atomics
  const float* pe;           // [32, N]
  int pe_fixed;              // EPI_RELU_PE: 0 = row m uses pe[m & 31]; 1 = every row uses pe[0] (caller points pe at one time step)
  float* dbias;              // EPI_ATOMIC: column sums of A over k (grad of the bias), or nullptr
  float mask_scale;
  int bf16;                  // gt_config.precision: 1 = the operands go through the matrix cores as bf16 (fp32 accumulate)
  DropArgs drop;
  // bf16 SHADOWS (precision = 1, round 4): A16 [M][K] / B16 [N][K], k contiguous -- bf16-rounded copies of A and of the weight (for a
  // dgrad: of its transpose, which turns the product into this same NT form), written by the tensors' producers / the per-step weight
  // shadow kernel.  With both present the product runs on gemm32h_kernel (gt_gemm32.h): half the bytes staged per flop, bitwise the
  // results of the fp32-source kernel, which rounds the same values at fragment assembly.  C16 (ldc16): optional bf16 copy of the
  // OUTPUT, for the next consumer.  nullptr: none.
  const uint16_t* A16; const uint16_t* B16; int lda16, ldb16;
  uint16_t* C16; int ldc16;
  const uint16_t* res16;     // EPI_MASK_NZ: the mask source as bf16 (row stride ldres), when its fp32 tensor is not stored
  uint16_t* kbits;           // the FFN activation's keep bits (gt_gemm32.h, gemm32_store_epilogue): written by EPI_RELU_DROP, read by EPI_MASK_NZ in
                             // place of res / res16 -- ring-tile kernels only (the host sets it when BOTH launches are on them: ffn_kbits)
  int as_dgrad;              // profiling label only: an NT product that IS a dgrad (B = a transposed weight copy)
  // gt_gemm64.h, LayerNorm-fused epilogues on 64x64 tiles: the row exchange region of the workspace (header: error word, launch serial,
  // ticket; then [M][N / 32 parts][2] tagged 8-byte granules) and the bound of its polling loop
  unsigned* rowx; int spin_max;
  int round16;               // EPI_RES_LN on the big tile at precision 2: the Linear output (acc + bias) is rounded to bf16 before dropout / residual / norm --
                             // the value the un-fused form stores in bf16 ahead of its LayerNorm pass (what torch.autocast hands on)
};

template <int ROWS, int COLS, int NT>
struct TileStage {
  static constexpr int CPR = COLS / 4;            // float4 chunks per row
  static constexpr int CH = ROWS * CPR;
  static constexpr int PER = (CH + NT - 1) / NT;
  static constexpr bool FAST_OK = (CH % NT) == 0;
  float4 v[PER];
  const float* zp;           // gt_zero_ptr() of the enclosing kernel
  uint32_t boff[PER];        // interior fast path: byte offset of chunk i from the (wave-uniform) slab origin
  // Interior tiles (whole tile in range, 16-byte aligned rows, full slabs) need none of the per-chunk bounds work below:
  // the offsets are computed once, the slab origin is a scalar, and each load is `global_load_dwordx4 v, v_off, s[base]`.
  // With one wave per SIMD (128x128 tiles) the ~150 VALU instructions per slab of the checked form are not hidden behind
  // anything -- they were 15-25 % of a slab's MFMA time.
  __device__ __forceinline__ void prep(int ld, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int ch = tid + i * NT;
      boff[i] = (uint32_t)((ch / CPR) * ld + (ch % CPR) * 4) * 4u;
    }
  }
  __device__ __forceinline__ void load_fast(const float* __restrict__ origin) {
#pragma unroll
    for (int i = 0; i < PER; ++i)
      v[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(origin) + boff[i]);
  }
  // Branch-free staging: every lane ALWAYS issues its loads (out-of-range chunks read gt_zero_page).  A guarded `if (ok) t = load` makes hipcc branch around each load
  // and wait vmcnt(0) per element -- one serialized L2 round trip per chunk (cdna_hip_programming.md 5,
  // trap (c)); measured here as ~3.7 us per 64-wide slab before this form.
  __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int r0, int c0, int rmax,
                                       int cmax, bool vec, int tid) {
    if (vec && (cmax & 3) == 0) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int ch = tid + i * NT;
        const int r = ch / CPR, c = (ch % CPR) * 4;
        const int gr = r0 + r, gc = c0 + c;
        const bool ok = (CH % NT == 0 || ch < CH) && gr < rmax && gc < cmax;
        const float* p = ok ? src + ((size_t)gr * ld + gc) : zp;
        v[i] = *reinterpret_cast<const float4*>(p);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int ch = tid + i * NT;
        const int r = ch / CPR, c = (ch % CPR) * 4;
        const int gr = r0 + r, gc = c0 + c;
        const bool okr = (CH % NT == 0 || ch < CH) && gr < rmax;
        const float* base = src + ((size_t)gr * ld + gc);
        const float* p0 = (okr && gc < cmax) ? base : zp;
        const float* p1 = (okr && gc + 1 < cmax) ? base + 1 : zp;
        const float* p2 = (okr && gc + 2 < cmax) ? base + 2 : zp;
        const float* p3 = (okr && gc + 3 < cmax) ? base + 3 : zp;
        v[i] = make_float4(*p0, *p1, *p2, *p3);
      }
    }
  }
  __device__ __forceinline__ void store(float* s, int str, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int ch = tid + i * NT;
      if (CH % NT == 0 || ch < CH) {
        int r = ch / CPR, c = (ch % CPR) * 4;
        *reinterpret_cast<float4*>(&s[r * str + c]) = v[i];
      }
    }
  }
  // bf16 operand path: the LDS image is ALWAYS [tile row][k] in bf16 (row stride str16 elements), whatever the source's
  // orientation, so the fragment of 8 consecutive k is one ds_read_b128.  Conversion happens here, on the way into LDS.
  //   source k-contiguous (this tile = [rows][BK]): chunk (r, k = c..c+3) -> one 8-byte store
  __device__ __forceinline__ void store_bf16_kc(uint16_t* s, int str16, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int ch = tid + i * NT;
      if (CH % NT == 0 || ch < CH) {
        int r = ch / CPR, c = (ch % CPR) * 4;
        uint2 pk;
        pk.x = (uint32_t)gt_f2bf(v[i].x) | ((uint32_t)gt_f2bf(v[i].y) << 16);
        pk.y = (uint32_t)gt_f2bf(v[i].z) | ((uint32_t)gt_f2bf(v[i].w) << 16);
        *reinterpret_cast<uint2*>(&s[r * str16 + c]) = pk;
      }
    }
  }
  //   source row-contiguous (this tile = [BK k-rows][rows]): chunk (k = r, rows c..c+3) -> transposed, four 2-byte stores
  __device__ __forceinline__ void store_bf16_tr(uint16_t* s, int str16, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int ch = tid + i * NT;
      if (CH % NT == 0 || ch < CH) {
        int r = ch / CPR, c = (ch % CPR) * 4;
        s[(c + 0) * str16 + r] = gt_f2bf(v[i].x);
        s[(c + 1) * str16 + r] = gt_f2bf(v[i].y);
        s[(c + 2) * str16 + r] = gt_f2bf(v[i].z);
        s[(c + 3) * str16 + r] = gt_f2bf(v[i].w);
      }
    }
  }
};

template <int WM, int WN, int TM, int TN, int BK_, bool AKM, bool BKM, int EPI, int PREC = 0>
struct GemmCfg {
  static constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = BK_, NT = WM * WN * 64;
  // bf16 operand path (PREC = 1): [tile row][k] bf16 images; row stride BK + 16 elements = (2 mod 4) 16-byte slots, the
  // conflict-free condition of ds_read_b128 derived below for the fp32 tiles
  static constexpr int STR16 = BK + 16;
  static_assert(PREC == 0 || (BK % 32 == 0), "bf16 MFMA contracts 32 k per instruction");
  // k-contiguous tiles are read with ds_read_b128, which the LDS services in four NON-contiguous 16-lane groups
  // ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md "LDS"), each mixing all 16 rows at two neighbouring k-offsets: the row
  // stride must be == 8 (mod 16) floats for the 16 slots of a group to be distinct.  BK+4 (an odd number of 16-byte slots)
  // made 7 of 8 pairs collide -- SQ_LDS_BANK_CONFLICT was 33-48 % of SQ_LDS_IDX_ACTIVE in the NT kernels.
  // Row-contiguous tiles (ds_read_b32, two 32-lane groups, banks mod 32) are conflict-free at +4.
  static constexpr int SA_STR = AKM ? BM + 4 : BK + 8, SA_ROWS = AKM ? BK : BM;
  static constexpr int SB_STR = BKM ? BN + 4 : BK + 8, SB_ROWS = BKM ? BK : BN;
  static constexpr int SA_SZ = PREC ? BM * STR16 / 2 : SA_ROWS * SA_STR, SB_SZ = PREC ? BN * STR16 / 2 : SB_ROWS * SB_STR;   // dwords
  static constexpr bool ROW = (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD);
  static constexpr int CSTR = BN + 4;
  static constexpr int NG = WM * WN * 4;                       // 16-lane row groups per workgroup
  static constexpr int MAIN_SZ = 2 * (SA_SZ + SB_SZ);
  // row epilogue: BM x CSTR staging; LN backward re-uses it for the [NG][CSTR] x 2 dgamma/dbeta partials
  static constexpr int EPI_ROWS = EPI == EPI_RES_LNBWD ? (BM > 2 * NG ? BM : 2 * NG) : BM;
  static constexpr int EPI_SZ = ROW ? EPI_ROWS * CSTR : 0;
  static constexpr int SMEM = MAIN_SZ > EPI_SZ ? MAIN_SZ : EPI_SZ;
};

__device__ static inline float gt_red16(float v) {
  v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
  return v;
}

// Row epilogues on a staged tile: sC (LDS, [BM][CSTR], CSTR = BN + 4) holds the raw accumulators of rows m0 .. m0 + BM - 1 (all BN
// columns: the workgroup owns whole rows); every 16-lane group owns one row at a time, lane l16 holds columns l16 + 16 i in registers
// and the LayerNorm statistics are 16-lane xor-shuffle sums.  Shared by gemm_body's row tiles and the ring-body row kernel
// (gt_gemm32.h).  NT threads, NG = NT / 16 row groups, BM % NG == 0.  smem: sC itself -- the dgamma / dbeta partials reuse it.
template <int BM, int BN, int NT, int EPI>
__device__ __forceinline__ void gemm_row_epilogue(const GemmArgs& g, const int m0, const int by, float* smem) {
  constexpr int CSTR = BN + 4, NG = NT / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const float* const zp = gt_zero_ptr();
  auto ldg = [zp](const float* p, size_t idx, bool ok) -> float { return *(ok ? p + idx : zp); };
  const uint32_t dkey = gt_drop_key(g.drop);
  float* sC = smem;
  constexpr int CPL = BN / 16;            // columns per lane of a 16-lane row group
  const int grp = wave * 4 + lg;
  const float invN = 1.0f / (float)g.N;
  float dg[EPI == EPI_RES_LNBWD ? CPL : 1], db[EPI == EPI_RES_LNBWD ? CPL : 1];
  if (EPI == EPI_RES_LNBWD) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) { dg[i] = 0.f; db[i] = 0.f; }
  }

  for (int rl = grp; rl < BM; rl += NG) {           // BM % NG == 0: the trip count is wave-uniform
    const int row = m0 + rl;
    const bool live = row < g.M;
    const size_t rowc = live ? row : 0;               // clamped row for the unconditional loads
    const float* zr = sC + rl * CSTR;
    float z[CPL], e1[CPL], e2[CPL], e3[CPL];
    if (EPI == EPI_RES_LN) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {                 // phase 1: all loads
        const int c = l16 + 16 * i;
        const bool okc = c < g.N;
        z[i] = zr[c];
        e1[i] = ldg(g.bias, c, okc);
        e2[i] = ldg(g.res, rowc * g.ldres + c, okc);
        e3[i] = ldg(g.gamma, c, okc);
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        z[i] = (c < g.N) ? (z[i] + e1[i]) * gt_drop_mul(g.drop, dkey, (uint32_t)(rowc * g.N + c)) + e2[i] : 0.f;
        s += z[i];
      }
      const float mean = gt_red16(s) * invN;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) { const int c = l16 + 16 * i; if (c < g.N) { const float d = z[i] - mean; q += d * d; } }
      const float rstd = 1.0f / sqrtf(gt_red16(q) * invN + GT_LN_EPS);
#pragma unroll
      for (int i = 0; i < CPL; ++i) e1[i] = ldg(g.beta, l16 + 16 * i, l16 + 16 * i < g.N);
      if (live) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int c = l16 + 16 * i;
          if (c < g.N) {
            const float xh = (z[i] - mean) * rstd;
            g.aux[(size_t)row * g.N + c] = xh;
            g.C[(size_t)row * g.ldc + c] = xh * e3[i] + e1[i];
          }
        }
        if (l16 == 0) g.aux2[row] = rstd;
      }
    } else {
      const float rs = g.rstd[rowc];
#pragma unroll
      for (int i = 0; i < CPL; ++i) {                 // phase 1: all loads
        const int c = l16 + 16 * i;
        const bool okc = live && c < g.N;
        z[i] = zr[c];
        e1[i] = ldg(g.res, rowc * g.ldres + c, okc && g.res != nullptr);
        e2[i] = ldg(g.xhat, rowc * g.N + c, okc);
        e3[i] = ldg(g.gamma, c, okc);
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        const float dy = (live && c < g.N) ? z[i] + e1[i] : 0.f;
        z[i] = dy;
        const float gdy = dy * e3[i];
        s1 += gdy; s2 += gdy * e2[i];
        dg[i] += dy * e2[i]; db[i] += dy;
      }
      const float m1 = gt_red16(s1) * invN, m2 = gt_red16(s2) * invN;
      if (live) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int c = l16 + 16 * i;
          if (c < g.N) {
            const float dz = rs * (z[i] * e3[i] - m1 - e2[i] * m2);
            g.C[(size_t)row * g.ldc + c] = dz;
            if (g.C2) g.C2[(size_t)row * g.ldc + c] = dz * gt_drop_mul(g.drop, dkey, (uint32_t)(row * g.N + c));
          }
        }
      }
    }
  }
  if (EPI == EPI_RES_LNBWD) {
    // dgamma/dbeta: reduce the NG row groups through LDS, then ONE atomic per column per workgroup
    __syncthreads();
    float* sG = smem;
    float* sBt = smem + NG * CSTR;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      sG[grp * CSTR + c] = dg[i];
      sBt[grp * CSTR + c] = db[i];
    }
    __syncthreads();
    for (int c = tid; c < g.N; c += NT) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int q = 0; q < NG; ++q) { a += sG[q * CSTR + c]; b += sBt[q * CSTR + c]; }
      if (g.ln_part) {       // 128+ workgroups adding into the same d addresses serialise at the atomic unit (~6 us):
        g.ln_part[((size_t)by * 2) * g.N + c] = a;             // store partials, ln_param_reduce_kernel sums them later
        g.ln_part[((size_t)by * 2 + 1) * g.N + c] = b;
      } else {
        atomicAdd(&g.dgamma[c], a);
        atomicAdd(&g.dbeta[c], b);
      }
    }
  }
}

// (round 6, defined in gt_gemm64.h beside the other geometries of the row exchange)
__device__ __forceinline__ uint32_t gemm_xln32_tag(const GemmArgs& g, const int m0, const int n0);
template <int EPI>
__device__ __forceinline__ void gemm_xln32_epilogue(const GemmArgs& g, const int m0, const int n0, const uint32_t tag0, float* smem);

template <int WM, int WN, int TM, int TN, int BK_, bool AKM, bool BKM, int EPI, int PREC = 0>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz, float* smem) {
  typedef GemmCfg<WM, WN, TM, TN, BK_, AKM, BKM, EPI, PREC> Cfg;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NT = Cfg::NT;
  constexpr int SA_STR = Cfg::SA_STR, SB_STR = Cfg::SB_STR, SA_SZ = Cfg::SA_SZ, SB_SZ = Cfg::SB_SZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l16 = lane & 15, lg = lane >> 4;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * g.k_chunk;
  const int kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;
  const int nk = (kend - kbeg + BK - 1) / BK;
  constexpr bool XLN = (EPI == EPI_RES_LN_X || EPI == EPI_RES_LNBWD_X);
  static_assert(!XLN || (BM == 32 && BN == 32 && NT == 256 && !AKM), "row-exchange epilogue: the 32x32 tile of 256 threads");
  uint32_t xtag = 0u;           // this launch's sequence number of the row exchange (read before anything is published)
  if constexpr (XLN) xtag = gemm_xln32_tag(g, m0, n0);

  const bool vecA = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool vecB = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* const zp = gt_zero_ptr();
  TileStage<Cfg::SA_ROWS, AKM ? BM : BK, NT> la;
  TileStage<Cfg::SB_ROWS, BKM ? BN : BK, NT> lb;
  la.zp = zp; lb.zp = zp;
  float bsum = 0.f;   // EPI_ATOMIC: bias-grad partial (column sum of the A slab)

  // wave-uniform: this workgroup's tile lies wholly inside A and B and every slab is full
  const bool fast = decltype(la)::FAST_OK && decltype(lb)::FAST_OK && vecA && vecB && nk > 0 && ((kend - kbeg) % BK) == 0 &&
                    m0 + BM <= g.M && n0 + BN <= g.N;

  auto main_loop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    if (FAST) { la.prep(g.lda, tid); lb.prep(g.ldb, tid); }
    auto load_tiles = [&](int k0) {
      if (FAST) {
        la.load_fast(AKM ? g.A + ((size_t)k0 * g.lda + m0) : g.A + ((size_t)m0 * g.lda + k0));
        lb.load_fast(BKM ? g.B + ((size_t)k0 * g.ldb + n0) : g.B + ((size_t)n0 * g.ldb + k0));
      } else {
        if (AKM) la.load(g.A, g.lda, k0, m0, kend, g.M, vecA, tid);
        else     la.load(g.A, g.lda, m0, k0, g.M, kend, vecA, tid);
        if (BKM) lb.load(g.B, g.ldb, k0, n0, kend, g.N, vecB, tid);
        else     lb.load(g.B, g.ldb, n0, k0, g.N, kend, vecB, tid);
      }
    };

    if (nk > 0) {
      load_tiles(kbeg);
      la.store(smem, SA_STR, tid);
      lb.store(smem + 2 * SA_SZ, SB_STR, tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const float* sA = smem + cur * SA_SZ;
      const float* sB = smem + 2 * SA_SZ + cur * SB_SZ;
      if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
      // (no tail skip: slabs are zero-filled beyond K; a branch here splits the MFMA block and hipcc then shuttles the
      //  accumulators VGPR<->AGPR around every few MFMAs, exposing the LDS latency each time)
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        if (GT_MIDSLAB_STORE && BK / 16 >= 2 && TM * TN >= 16 && kk == BK / 32 && kt + 1 < nk) {
          // mid-slab hand-over: the next slab goes into the other LDS buffer while half of this slab's MFMAs are still to
          // come (that buffer was last read before the previous barrier), so the barrier below waits on nothing
          la.store(smem + (cur ^ 1) * SA_SZ, SA_STR, tid);
          lb.store(smem + 2 * SA_SZ + (cur ^ 1) * SB_SZ, SB_STR, tid);
        }
        float af[TM][4], bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = (wm * TM + i) * 16 + l16;
          if (!AKM) {
            float4 t = *reinterpret_cast<const float4*>(&sA[row * SA_STR + kk * 16 + 4 * lg]);
            af[i][0] = t.x; af[i][1] = t.y; af[i][2] = t.z; af[i][3] = t.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) af[i][j] = sA[(kk * 16 + 4 * lg + j) * SA_STR + row];
          }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int col = (wn * TN + i) * 16 + l16;
          if (!BKM) {
            float4 t = *reinterpret_cast<const float4*>(&sB[col * SB_STR + kk * 16 + 4 * lg]);
            bf[i][0] = t.x; bf[i][1] = t.y; bf[i][2] = t.z; bf[i][3] = t.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[i][j] = sB[(kk * 16 + 4 * lg + j) * SB_STR + col];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)      // D = Btile Atile^T except for the atomic epilogue: see "accumulator layout"
              acc[a][b] = (EPI != EPI_ATOMIC) ? GT_MFMA16(bf[b][j], af[a][j], acc[a][b]) : GT_MFMA16(af[a][j], bf[b][j], acc[a][b]);
      }

      if (EPI == EPI_ATOMIC && AKM) {
        if (g.dbias != nullptr && bx == 0 && tid < BM) {
#pragma unroll 8
          for (int kk = 0; kk < BK; ++kk) bsum += sA[kk * SA_STR + tid];
        }
      }
      if (kt + 1 < nk && !(GT_MIDSLAB_STORE && BK / 16 >= 2 && TM * TN >= 16)) {
        la.store(smem + (cur ^ 1) * SA_SZ, SA_STR, tid);
        lb.store(smem + 2 * SA_SZ + (cur ^ 1) * SB_SZ, SB_STR, tid);
      }
      __syncthreads();
    }
  };
  // bf16 operands: same staging loads (fp32 from global), conversion on the way into LDS, one ds_read_b128 per fragment,
  // v_mfma_f32_16x16x32_bf16.  The accumulator layout equals the fp32 form's, so every epilogue below is shared.
  auto main_loop_bf16 = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr int STR16 = Cfg::STR16, SA16 = BM * STR16, SB16 = BN * STR16;
    uint16_t* const s16 = reinterpret_cast<uint16_t*>(smem);
    if (FAST) { la.prep(g.lda, tid); lb.prep(g.ldb, tid); }
    auto load_tiles = [&](int k0) {
      if (FAST) {
        la.load_fast(AKM ? g.A + ((size_t)k0 * g.lda + m0) : g.A + ((size_t)m0 * g.lda + k0));
        lb.load_fast(BKM ? g.B + ((size_t)k0 * g.ldb + n0) : g.B + ((size_t)n0 * g.ldb + k0));
      } else {
        if (AKM) la.load(g.A, g.lda, k0, m0, kend, g.M, vecA, tid);
        else     la.load(g.A, g.lda, m0, k0, g.M, kend, vecA, tid);
        if (BKM) lb.load(g.B, g.ldb, k0, n0, kend, g.N, vecB, tid);
        else     lb.load(g.B, g.ldb, n0, k0, g.N, kend, vecB, tid);
      }
    };
    auto store_tiles = [&](int buf) {
      uint16_t* a = s16 + buf * SA16;
      uint16_t* b = s16 + 2 * SA16 + buf * SB16;
      if (AKM) la.store_bf16_tr(a, STR16, tid); else la.store_bf16_kc(a, STR16, tid);
      if (BKM) lb.store_bf16_tr(b, STR16, tid); else lb.store_bf16_kc(b, STR16, tid);
    };
    if (nk > 0) { load_tiles(kbeg); store_tiles(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const uint16_t* sA = s16 + cur * SA16;
      const uint16_t* sB = s16 + 2 * SA16 + cur * SB16;
      if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        bf16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const bf16x8*>(&sA[((wm * TM + i) * 16 + l16) * STR16 + kk * 32 + 8 * lg]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
          bf[i] = *reinterpret_cast<const bf16x8*>(&sB[((wn * TN + i) * 16 + l16) * STR16 + kk * 32 + 8 * lg]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)        // transposed tile except for the atomic epilogue, as in the fp32 loop
            acc[a][b] = (EPI != EPI_ATOMIC) ? GT_MFMA16_BF16(bf[b], af[a], acc[a][b]) : GT_MFMA16_BF16(af[a], bf[b], acc[a][b]);
      }
      if (EPI == EPI_ATOMIC && AKM) {           // bias gradient: column sums of the staged (bf16) dY slab
        if (g.dbias != nullptr && bx == 0 && tid < BM) {
#pragma unroll 8
          for (int kk = 0; kk < BK; ++kk) bsum += gt_bf2f(sA[tid * STR16 + kk]);
        }
      }
      if (kt + 1 < nk) store_tiles(cur ^ 1);
      __syncthreads();
    }
  };
  if constexpr (PREC == 1) {
    if (fast) main_loop_bf16(std::true_type{}); else main_loop_bf16(std::false_type{});
  } else {
    if (fast) main_loop(std::true_type{}); else main_loop(std::false_type{});
  }

  // ------------------------------------------------------------------------------- epilogues
  // Two-phase everywhere: (1) every global input of the epilogue is loaded UNCONDITIONALLY (address-select
  // against gt_zero_page for out-of-range elements) into registers, so all loads are in flight together;
  // (2) compute + predicated stores.  Guarded per-element loads would serialise one L2 round trip each.
  const uint32_t dkey = gt_drop_key(g.drop);
  auto ldg = [zp](const float* p, size_t idx, bool ok) -> float { return *(ok ? p + idx : zp); };

#ifdef GT_BENCH_NOEPI
  // tools/ubench/gemm_bench.hip only: main loop alone (the accumulators stay live through a store that never happens)
  if (EPI == EPI_STORE && g.mask_scale == 12345.f) {
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) t += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (t == 123.456f) g.C[0] = t;
    return;
  }
#endif
  if constexpr (XLN) {
    // the tile's raw accumulators -> LDS [32][36] (the main loop's final barrier has passed), then the exchange epilogue
    *reinterpret_cast<float4*>(&smem[(wm * 16 + l16) * 36 + wn * 16 + 4 * lg]) = make_float4(acc[0][0][0], acc[0][0][1], acc[0][0][2], acc[0][0][3]);
    __syncthreads();
    gemm_xln32_epilogue<EPI>(g, m0, n0, xtag, smem);
    return;
  }
  if (!Cfg::ROW) {
    if (EPI == EPI_ATOMIC && AKM) {
      if (g.dbias != nullptr && bx == 0 && tid < BM && m0 + tid < g.M) atomicAdd(&g.dbias[m0 + tid], bsum);
    }
    // Accumulator layout.  The MFMAs compute the TRANSPOSED tile (first operand = B fragment, second = A fragment), so lane
    // (l16, lg) holds, in the 4 registers of tile (a, b), output row  m0 + (wm TM + a) 16 + l16  and the four CONSECUTIVE
    // columns  n0 + (wn TN + b) 16 + 4 lg + r:  one 16-byte store per tile per lane (64-byte row segments per 4 lanes)
    // instead of four 4-byte stores -- the untransposed form spent 12 % of a 128x128 GEMM in its epilogue.
    // EPI_ATOMIC keeps the untransposed tile (rows 4 lg + r, column l16): one atomic instruction then covers 4 rows x 64
    // contiguous bytes; transposed it would touch 16 rows x 4 scattered dwords (measured: weight gradients 47 -> 78 us).
    constexpr bool TR = (EPI != EPI_ATOMIC);
    constexpr bool NEED_R1 = (EPI == EPI_STORE || EPI == EPI_RELU_PE || EPI == EPI_MASK_NZ || EPI == EPI_ADD_RELUMASK_DROP);
    constexpr bool NEED_R2 = (EPI == EPI_ADD_RELUMASK_DROP);
    constexpr bool NEED_BIAS = (EPI == EPI_STORE || EPI == EPI_RELU_PE || EPI == EPI_RELU_DROP || EPI == EPI_HEADS);
    const bool need_r1 = NEED_R1 && (EPI != EPI_STORE || g.accumulate != 0);      // wave-uniform: no dummy loads for a plain store
    const float* r1src = EPI == EPI_STORE ? g.C : EPI == EPI_RELU_PE ? g.pe : g.res;
    const int r1ld = EPI == EPI_STORE ? g.ldc : EPI == EPI_RELU_PE ? g.N : g.ldres;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // 16-byte path: every row start and every 4-column group is 16-byte aligned in all tensors the epilogue touches
    const bool vec = EPI != EPI_ATOMIC && EPI != EPI_HEADS && (g.N & 3) == 0 && (g.ldc & 3) == 0 && al16(g.C) &&
                     (!NEED_BIAS || g.bias == nullptr || al16(g.bias)) && (!need_r1 || ((r1ld & 3) == 0 && al16(r1src))) &&
                     (!NEED_R2 || al16(g.aux_in)) && (EPI != EPI_RELU_PE || al16(g.aux));
    auto row_of = [&](int a) { return m0 + (wm * TM + a) * 16 + (TR ? l16 : 4 * lg); };     // + r if !TR
    auto col_of = [&](int b) { return n0 + (wn * TN + b) * 16 + (TR ? 4 * lg : l16); };     // + r if TR
    float bia[TN][4], r1[NEED_R1 ? TM : 1][NEED_R1 ? TN : 1][4], r2[NEED_R2 ? TM : 1][NEED_R2 ? TN : 1][4];
    if (NEED_R1 && !need_r1) {                    // plain store: nothing to read back
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) r1[a][b][r] = 0.f;
    }
    // ---- phase 1: every global input of the epilogue, unconditionally (address-select against the zero page)
    if (vec) {
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = col_of(b);
        const float4 t = *reinterpret_cast<const float4*>((NEED_BIAS && g.bias != nullptr && col < g.N) ? g.bias + col : zp);
        bia[b][0] = t.x; bia[b][1] = t.y; bia[b][2] = t.z; bia[b][3] = t.w;
      }
      if (NEED_R1 && need_r1) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const int row = row_of(a), col = col_of(b);
            const bool ok = row < g.M && col < g.N;
            const int rrow = EPI == EPI_RELU_PE ? (g.pe_fixed ? 0 : (row & 31)) : row;
            const float4 t = *reinterpret_cast<const float4*>(ok ? r1src + ((size_t)rrow * r1ld + col) : zp);
            r1[a][b][0] = t.x; r1[a][b][1] = t.y; r1[a][b][2] = t.z; r1[a][b][3] = t.w;
            if (NEED_R2) {
              const float4 u = *reinterpret_cast<const float4*>(ok ? g.aux_in + ((size_t)row * g.N + col) : zp);
              r2[a][b][0] = u.x; r2[a][b][1] = u.y; r2[a][b][2] = u.z; r2[a][b][3] = u.w;
            }
          }
      }
    } else {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = col_of(b) + (TR ? r : 0);
          bia[b][r] = NEED_BIAS ? ldg(g.bias, col, g.bias != nullptr && col < g.N) : 0.f;
        }
      if (NEED_R1 && need_r1) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = row_of(a) + (TR ? 0 : r), col = col_of(b) + (TR ? r : 0);
              const bool ok = row < g.M && col < g.N;
              const int rrow = EPI == EPI_RELU_PE ? (g.pe_fixed ? 0 : (row & 31)) : row;
              r1[a][b][r] = ldg(r1src, (size_t)rrow * r1ld + col, ok);
              if (NEED_R2) r2[a][b][r] = ldg(g.aux_in, (size_t)row * g.N + col, ok);
            }
      }
    }
    // ---- phase 2: compute + predicated stores
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int row0 = row_of(a), col0 = col_of(b);
        float o[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + (TR ? 0 : r), col = col0 + (TR ? r : 0);
          float v = acc[a][b][r];
          const uint32_t didx = (uint32_t)(row * g.N + col);
          o2[r] = 0.f;
          if (EPI == EPI_STORE) {
            v = v + bia[b][r] + r1[a][b][r];
          } else if (EPI == EPI_RELU_PE) {
            v += bia[b][r];
            o2[r] = v;                                                   // aux = pre-activation
            v = (fmaxf(v, 0.f) + r1[a][b][r]) * gt_drop_mul(g.drop, dkey, didx);
          } else if (EPI == EPI_RELU_DROP) {
            v = fmaxf(v + bia[b][r], 0.f) * gt_drop_mul(g.drop, dkey, didx);
          } else if (EPI == EPI_HEADS) {
            v += bia[b][r];
            if (col >= 2 * GT_VOICES) v = 0.5f * tanhf(v);
            else if (col >= GT_VOICES) v = gt_sigmoid(v);
          } else if (EPI == EPI_MASK_NZ) {
            v = (r1[a][b][r] != 0.f) ? v * g.mask_scale : 0.f;
          } else if (EPI == EPI_ADD_RELUMASK_DROP) {
            v = (v + r1[a][b][r]) * gt_drop_mul(g.drop, dkey, didx);
            v = (r2[a][b][r] > 0.f) ? v : 0.f;
          }
          o[r] = v;
        }
        if (vec) {
          if (row0 < g.M && col0 < g.N) {
            *reinterpret_cast<float4*>(&g.C[(size_t)row0 * g.ldc + col0]) = make_float4(o[0], o[1], o[2], o[3]);
            if (EPI == EPI_RELU_PE) *reinterpret_cast<float4*>(&g.aux[(size_t)row0 * g.N + col0]) = make_float4(o2[0], o2[1], o2[2], o2[3]);
            if (EPI == EPI_RELU_PE && g.C16 != nullptr) {                 // bf16 shadow of the input layer's output (operand of layer 0's in-proj and of its weight gradient)
              uint2 pk;
              pk.x = (uint32_t)gt_f2bf(o[0]) | ((uint32_t)gt_f2bf(o[1]) << 16);
              pk.y = (uint32_t)gt_f2bf(o[2]) | ((uint32_t)gt_f2bf(o[3]) << 16);
              *reinterpret_cast<uint2*>(g.C16 + (size_t)row0 * g.ldc16 + col0) = pk;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + (TR ? 0 : r), col = col0 + (TR ? r : 0);
            if (row < g.M && col < g.N) {
              const size_t ci = (size_t)row * g.ldc + col;
              if (EPI == EPI_ATOMIC) atomicAdd(&g.C[ci], o[r]);
              else g.C[ci] = o[r];
              if (EPI == EPI_RELU_PE) g.aux[(size_t)row * g.N + col] = o2[r];
              if (EPI == EPI_RELU_PE && g.C16 != nullptr) g.C16[(size_t)row * g.ldc16 + col] = gt_f2bf(o[r]);
            }
          }
        }
      }
    return;
  }

  // Row epilogues: stage the raw BM x BN accumulators in LDS (the main loop's final barrier has passed),
  // then every 16-lane group owns one row at a time: lane l16 holds columns l16 + 16 i in registers and the
  // LayerNorm statistics are 16-lane xor-shuffle sums.
  constexpr int CSTR = Cfg::CSTR, NG = Cfg::NG;
  float* sC = smem;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)   // transposed accumulator layout (above): 4 consecutive columns of one row per lane
      *reinterpret_cast<float4*>(&sC[((wm * TM + a) * 16 + l16) * CSTR + (wn * TN + b) * 16 + 4 * lg]) =
          make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
  __syncthreads();

  gemm_row_epilogue<BM, BN, NT, EPI>(g, m0, by, smem);
}

// 128x128 tiles: ask for two waves per SIMD (<= 256 registers) so that two workgroups share a CU and one's prologue, barriers
// and epilogue hide behind the other's MFMAs (unconstrained, hipcc spends 320 registers and only one workgroup fits)
#ifndef GT_T128_WAVES
#define GT_T128_WAVES 2
#endif
template <int WM, int WN, int TM, int TN, int BK_, bool AKM, bool BKM, int EPI, int PREC = 0>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && TM * TN >= 16 && TN == 4 && EPI != EPI_ADD_RELUMASK_DROP && EPI != EPI_RELU_PE)
                                              ? GT_T128_WAVES : 1) void gemm_kernel(GemmArgs g) {
  // (the two input-layer epilogues hold a second and third 64-register operand set next to the accumulators: under the
  //  256-register cap of two waves per SIMD they spilled 380 / 12 bytes per lane -- they run once per step, one wave is fine)
  __shared__ __attribute__((aligned(16))) float smem[GemmCfg<WM, WN, TM, TN, BK_, AKM, BKM, EPI, PREC>::SMEM];
#ifndef GT_NO_XCD_REMAP
  if (gridDim.z == 1) {
    // XCD-aware placement (as in wgrad_group_kernel): workgroups are dealt round-robin over the 8 XCDs in dispatch order
    // (x fastest), so the tiles of one A row-panel would land in 8 different L2s and the panel would cross the fabric 8
    // times -- the waves of the 128x128 kernel spent 71 % of their cycles in s_waitcnt on exactly those loads.  Give each
    // XCD a contiguous range of the x-fastest tile order instead.  Placement changes speed only, never results.
    const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
    const int xcd = lin & 7, q = nb >> 3, r = nb & 7;
    const int b = xcd * q + (xcd < r ? xcd : r) + (lin >> 3);
    gemm_body<WM, WN, TM, TN, BK_, AKM, BKM, EPI, PREC>(g, b % gx, b / gx, 0, smem);
    return;
  }
#endif
  gemm_body<WM, WN, TM, TN, BK_, AKM, BKM, EPI, PREC>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Grouped launch: up to GT_GROUP_MAX independent GEMMs of one kind in ONE dispatch (all weight gradients of a
// layer).  A kernel boundary costs ~4 us on this machine whatever the kernel does, and each wgrad alone is too
// small to fill 256 CUs; together they do.  Workgroup b serves problem i with start[i] <= b < start[i+1].
#define GT_GROUP_MAX 16
struct GemmGroup {
  int n;
  int start[GT_GROUP_MAX + 1];
  int gx[GT_GROUP_MAX], gy[GT_GROUP_MAX];
  GemmArgs p[GT_GROUP_MAX];
};
// weight-gradient group ("TN", fp32 atomics).  TM = 1: 32x32 tiles -- small problems, where the number of workgroups
// matters most (64x64 tiles were measured slower at M = 2048, twice: fewer workgroups -> longer serial slab chains).
// TM = 2: 64x64 tiles -- large problems (d_model 512, thousands of tokens), where 32x32 tiles are bound by staged bytes
// and atomics per flop (38 % of the C4 step before the split).  The two sizes go out as SEPARATE launches: sharing one,
// the 70 KB of LDS of the large tiles halved the occupancy of the small ones.
#ifndef GT_WGRAD_T32_BK
#define GT_WGRAD_T32_BK 32      /* 18 KB of LDS -> 8 resident workgroups per CU: these short, staging-bound chains want occupancy (64: +3 % step time at C2, 16 and 128 worse) */
#endif
#ifndef GT_WGRAD_T128_BK
#define GT_WGRAD_T128_BK 16      /* 34 KB of LDS: more resident workgroups (32: C3 +2.5 %, C4 bs512 +3.4 % step time) */
#endif
#ifndef GT_WGRAD_T64_BK
#define GT_WGRAD_T64_BK 64
#endif
#ifndef GT_WGRAD_T64_MIN
#define GT_WGRAD_T64_MIN 512
#endif
template <int TM, int PREC = 0>
__global__ __launch_bounds__(256, TM == 4 ? GT_T128_WAVES : 1) void wgrad_group_kernel(GemmGroup grp) {
  // token slab per step; the bf16 MFMA contracts 32 tokens per instruction, so its slabs are at least that long
  constexpr int BK0 = TM == 4 ? GT_WGRAD_T128_BK : TM == 2 ? GT_WGRAD_T64_BK : GT_WGRAD_T32_BK;
  constexpr int BK = (PREC == 1 && BK0 < 32) ? 32 : BK0;
  typedef GemmCfg<2, 2, TM, TM, BK, true, true, EPI_ATOMIC, PREC> C;
  __shared__ __attribute__((aligned(16))) float smem[C::SMEM];
  // XCD-aware placement: hardware deals workgroups round-robin over the 8 XCDs (b and b+8 share one), each with its own
  // L2.  Give every XCD a CONTIGUOUS range of the logical tile space (tiles of one problem / one token chunk share their
  // dY and X slabs), so each slab is pulled through the fabric once instead of once per XCD.  Placement only changes
  // speed, never results.
  const int nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, r = nb & 7;
  const int b = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
  int i = 0;
  while (i + 1 < grp.n && b >= grp.start[i + 1]) ++i;
  const int local = b - grp.start[i];
  const int bx = local % grp.gx[i], t = local / grp.gx[i];
  gemm_body<2, 2, TM, TM, BK, true, true, EPI_ATOMIC, PREC>(grp.p[i], bx, t % grp.gy[i], t / grp.gy[i], smem);
}

// --------------------------------------------------------------------------------- host dispatch
template <bool BKM, int EPI>
static inline const char* gemm_label() {
  return EPI == EPI_ATOMIC ? "gemm_wgrad" : (EPI == EPI_RES_LN || EPI == EPI_RES_LN_X) ? "gemm_fwd_res_ln" : (EPI == EPI_RES_LNBWD || EPI == EPI_RES_LNBWD_X) ? "gemm_dgrad_lnbwd"
       : EPI == EPI_RELU_PE ? "gemm_fwd_input" : EPI == EPI_RELU_DROP ? "gemm_fwd_ffn1" : EPI == EPI_HEADS ? "gemm_fwd_heads"
       : EPI == EPI_MASK_NZ ? "gemm_dgrad_ffn2" : EPI == EPI_ADD_RELUMASK_DROP ? "gemm_dgrad_input"
       : BKM ? "gemm_dgrad" : "gemm_fwd_bias";
}
template <int WM, int WN, int TM, int TN, int BK, bool AKM, bool BKM, int EPI, int PREC = 0>
static inline void gemm_launch_cfg(const GemmArgs& g, int splitk, hipStream_t s) {
  typedef GemmCfg<WM, WN, TM, TN, BK, AKM, BKM, EPI, PREC> Cfg;
  static_assert(!Cfg::ROW || (Cfg::BM % Cfg::NG) == 0, "row epilogue: BM must be a multiple of the row-group count");
  gt_prof_tag((g.as_dgrad && EPI == EPI_STORE) ? "gemm_dgrad" : gemm_label<BKM, EPI>(), 2.0 * g.M * g.N * g.K,
              4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
  dim3 grid((g.N + Cfg::BN - 1) / Cfg::BN, (g.M + Cfg::BM - 1) / Cfg::BM, splitk);
  gt_launch(gemm_kernel<WM, WN, TM, TN, BK, AKM, BKM, EPI, PREC>, grid, dim3(Cfg::NT), s, g);
}

// wgrad: the output (N_w x K_w) is small, the contraction (tokens) is long -> split it over z so that the
// problem yields about `target` 32x32-tile workgroups.  Sets g.k_chunk, returns the split count.
// Deterministic mode (gt_set_deterministic / GT_DETERMINISTIC=1): ONE workgroup per output tile walks all tokens, so every
// gradient element (and bias-gradient element) has exactly one contributor and the fp32 atomics add onto a known value in a fixed
// order -- gradients, and with them the whole training run, repeat bit for bit.  Costs the small shapes their token parallelism.
#ifndef GT_WGRAD128_MIN_CHUNK
#define GT_WGRAD128_MIN_CHUNK 512
#endif
#ifndef GT_WGRAD128H_MIN_CHUNK
#define GT_WGRAD128H_MIN_CHUNK 1024     /* bf16 operands: measured at 2048 tokens (C5 bs 64): chunks >= 512 / 1024 / 2048 -> 0.958 / 0.946 / 0.984 ms (precision 2:
                                           0.952 / 0.927 / 0.968); fp32 (C4 bs 64): 1.494 / 1.508 / 1.572 -- stays at 512 */
#endif
static inline long gt_env_long(const char* name, long dflt);
static int g_deterministic = -1;
static inline bool gt_deterministic() {
  if (g_deterministic < 0) { const char* e = getenv("GT_DETERMINISTIC"); g_deterministic = (e && e[0] == '1') ? 1 : 0; }
  return g_deterministic != 0;
}
static inline int wgrad_split(GemmArgs& g, long target, int tile = 32) {
  if (gt_deterministic()) target = 1;
  const long tiles = (long)((g.M + tile - 1) / tile) * ((g.N + tile - 1) / tile);
  long want = (target + tiles - 1) / tiles;
  // (128x128 tiles: token chunks of at least GT_WGRAD128_MIN_CHUNK -- every chunk ends in 64 KB of fp32 atomics per tile)
  static const long c128 = gt_env_long("GT_WGRAD128_MIN_CHUNK", GT_WGRAD128_MIN_CHUNK), c128h = gt_env_long("GT_WGRAD128H_MIN_CHUNK", GT_WGRAD128H_MIN_CHUNK);
  const long ch128 = g.bf16 ? c128h : c128;
  const long maxs = tile >= 128 ? (g.K + ch128 - 1) / ch128 : tile >= 64 ? (g.K + 255) / 256 : (g.K + 127) / 128;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  int chunk = (int)((g.K + want - 1) / want);
  chunk = (chunk + 63) / 64 * 64;
  g.k_chunk = chunk;
  return (g.K + chunk - 1) / chunk;
}

// all weight gradients queued since the last flush go out as (at most) three grouped dispatches, one per tile size.
// GT_WGRAD_T128_MIN counts ONE problem's 128x128-tile workgroups; a launch groups the 4-6 problems of a layer, so 32 per
// problem already fills the chip (A/B on one box: 512 -> 32 took C3 from 6.92 to 6.22 ms; 16 pulls C2's M=2048 problems in
// and costs the headline shape 30 %).
// workgroups one problem is split into (over the token dimension); a dispatch groups every problem of the step
#ifndef GT_WGRAD_SPLIT_SMALL
#define GT_WGRAD_SPLIT_SMALL 512
#endif
#ifndef GT_WGRAD_SPLIT_BIG
#define GT_WGRAD_SPLIT_BIG 1024
#endif
#ifndef GT_WGRAD_T128_MIN
#define GT_WGRAD_T128_MIN 32
#endif
#ifndef GT_WGRAD32_SPLIT
#define GT_WGRAD32_SPLIT 1024   /* A/B on C4 bs512: 64 / 128 / 256 / 1024 -> 10.6 / 10.9 / 10.5 / 9.76 ms: many short chunks balance the grouped launch */
#endif
#ifndef GT_WGRAD32_CHUNK
#define GT_WGRAD32_CHUNK 2048
#endif
#ifndef GT_WGRAD32_UNIFORM_MIN
#define GT_WGRAD32_UNIFORM_MIN 16384
#endif
#define GT_WG_CLASSES 6
struct WgradBatch {
  // [0]: 32x32-tile problems, [1]: 64x64, [2]: 128x128, [3]: 128x128 on the big-tile body (gt_gemm32.h); [4] / [5]: the same body with both
  // operands / only dY staged from their bf16 shadows (precision = 1; GemmArgs::A16 / B16)
  GemmGroup grp[GT_WG_CLASSES];
  double flops[GT_WG_CLASSES], bytes[GT_WG_CLASSES];
  int bf16;                    // every problem of a batch shares the step's precision
  WgradBatch() : bf16(0) { for (int k = 0; k < GT_WG_CLASSES; ++k) { grp[k].n = 0; grp[k].start[0] = 0; flops[k] = bytes[k] = 0; } }
  bool empty() const { for (int k = 0; k < GT_WG_CLASSES; ++k) if (grp[k].n) return false; return true; }
};
static inline bool wgrad32_ok(const GemmArgs& g);
template <int PREC, bool SA16, bool SB16> __global__ __launch_bounds__(256, 2) void wgrad32_group_kernel(GemmGroup grp);
__global__ void wgrad32t_group_kernel(GemmGroup grp);
#ifndef GT_WGRAD32T
#define GT_WGRAD32T 1           /* class 4 (both operands bf16) on the transposed-LDS-read kernel; 0: widened into the fp32 image */
#endif
static inline void wgrad_flush_one(WgradBatch& wb, int k, hipStream_t s) {
  GemmGroup& G = wb.grp[k];
  if (G.n == 0) return;
  gt_prof_tag("gemm_wgrad", wb.flops[k], wb.bytes[k]);
  if (k == 4 && GT_WGRAD32T) gt_launch(wgrad32t_group_kernel, dim3(G.start[G.n]), dim3(256), s, G);
  else if (k == 4) gt_launch(wgrad32_group_kernel<1, true, true>, dim3(G.start[G.n]), dim3(256), s, G);
  else if (k == 5) gt_launch(wgrad32_group_kernel<1, true, false>, dim3(G.start[G.n]), dim3(256), s, G);
  else if (k == 3) {
    if (wb.bf16) gt_launch(wgrad32_group_kernel<1, false, false>, dim3(G.start[G.n]), dim3(256), s, G);
    else         gt_launch(wgrad32_group_kernel<0, false, false>, dim3(G.start[G.n]), dim3(256), s, G);
  } else if (wb.bf16) {
    if (k == 0)      gt_launch(wgrad_group_kernel<1, 1>, dim3(G.start[G.n]), dim3(256), s, G);
    else if (k == 1) gt_launch(wgrad_group_kernel<2, 1>, dim3(G.start[G.n]), dim3(256), s, G);
    else             gt_launch(wgrad_group_kernel<4, 1>, dim3(G.start[G.n]), dim3(256), s, G);
  } else {
    if (k == 0)      gt_launch(wgrad_group_kernel<1>, dim3(G.start[G.n]), dim3(256), s, G);
    else if (k == 1) gt_launch(wgrad_group_kernel<2>, dim3(G.start[G.n]), dim3(256), s, G);
    else             gt_launch(wgrad_group_kernel<4>, dim3(G.start[G.n]), dim3(256), s, G);
  }
  G.n = 0; wb.flops[k] = wb.bytes[k] = 0;
}
static inline void wgrad_flush(WgradBatch& wb, hipStream_t s) { for (int k = 0; k < GT_WG_CLASSES; ++k) wgrad_flush_one(wb, k, s); }
static inline void wgrad_queue(WgradBatch& wb, GemmArgs g, hipStream_t s) {
  // tile size by how many workgroups the problem still yields: 128x128 over >= 512-token chunks, else 64x64 over >= 256-token
  // chunks, else 32x32 (C2 at bs 64: 16 64x64-tiles x 8 chunks -> stays 32x32)
  const long t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64), t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
  int cls = (g.M >= 128 && g.N >= 128 && t128 * ((g.K + 511) / 512) >= GT_WGRAD_T128_MIN) ? 2
          : (g.M >= 64 && g.N >= 64 && t64 * ((g.K + 255) / 256) >= GT_WGRAD_T64_MIN) ? 1 : 0;
  const int tile = 32 << cls;
  if (cls == 2 && wgrad32_ok(g)) cls = 3;             // interior-only 128x128 problems: the prefetch-ring body
  if (cls == 3 && g.bf16 && g.A16 != nullptr && (g.lda16 & 7) == 0 && (reinterpret_cast<uintptr_t>(g.A16) & 15) == 0) {
    const bool b16 = g.B16 != nullptr && (g.ldb16 & 7) == 0 && (reinterpret_cast<uintptr_t>(g.B16) & 15) == 0;
    cls = b16 ? 4 : 5;                                 // ... staged from the operands' bf16 shadows
  }
  wb.bf16 = g.bf16;
  if (wb.grp[cls].n == GT_GROUP_MAX) wgrad_flush_one(wb, cls, s);
  // (big-tile body: fewer, longer token chunks -- every chunk ends in 64 KB of fp32 atomics per tile, and the chip adds
  //  ~1.3 TB/s of atomic bytes at most; a grouped launch has 16 problems' worth of workgroups anyway)
  // From GT_WGRAD32_UNIFORM_MIN tokens up every big-tile problem is cut into the SAME token-chunk length (GT_WGRAD32_CHUNK): all
  // workgroups of the grouped launch are then equal (no straggling problem) and every gradient tile takes 8 rounds of fp32 atomics
  // instead of 22-32 -- at 16384 tokens the per-problem target count had the d_model-512 step add ~1 GB of atomic bytes against
  // the chip's ~1.3 TB/s (C4 bs512: 9.79 -> 9.28 ms; at 2048 / 8192 tokens the shorter uneven chunks stay ahead: A/B in DESIGN 3a)
  int splitk;
  if (cls >= 3 && !gt_deterministic() && g.K >= GT_WGRAD32_UNIFORM_MIN) { g.k_chunk = GT_WGRAD32_CHUNK; splitk = (g.K + g.k_chunk - 1) / g.k_chunk; }
  else splitk = wgrad_split(g, cls >= 3 ? GT_WGRAD32_SPLIT : cls ? GT_WGRAD_SPLIT_BIG : GT_WGRAD_SPLIT_SMALL, tile);
  GemmGroup& G = wb.grp[cls];
  const int i = G.n++;
  G.p[i] = g;
  G.gx[i] = (g.N + tile - 1) / tile;
  G.gy[i] = (g.M + tile - 1) / tile;
  G.start[i + 1] = G.start[i] + G.gx[i] * G.gy[i] * splitk;
  wb.flops[cls] += 2.0 * g.M * g.N * g.K;
  wb.bytes[cls] += 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
}

#ifndef GT_T64_MIN
#define GT_T64_MIN 512
#endif
#ifndef GT_T32_BK
#define GT_T32_BK 64
#endif
#ifndef GT_T64_BK
#define GT_T64_BK 32
#endif
#ifndef GT_T128_MIN
#define GT_T128_MIN 512
#endif
// big-tile kernel of gt_gemm32.h (v_mfma_f32_32x32x2_f32, two-deep prefetch ring): interior-only large problems
static inline bool gemm32_ok(const GemmArgs& g, int epi, bool bkm);
template <bool BKM, int EPI>
static inline void gemm32_launch(const GemmArgs& g, hipStream_t s);
static inline bool gemm32h_ok(const GemmArgs& g, int epi);
template <bool BKM, int EPI>
static inline void gemm32h_launch(const GemmArgs& g, hipStream_t s);
#ifndef GT_T128_BIG_MIN
#define GT_T128_BIG_MIN 192     /* 128x128 tiles of the big kernel from this many workgroups (d512 QKV at 2048 tokens: 16 x 12) */
#endif
#ifndef GT_T128H_MIN
#define GT_T128H_MIN 64         /* ... the bf16-SOURCE kernel (operand shadows, precision 1) already from 64: at the bf16 MFMA rate a 128x128 tile is short, and
                                   the bf16-only storage it comes with pays by itself -- C5 at 64 sequences per GPU 1.282 -> 1.206 ms (the fp32 body at 64
                                   tiles: C4 bs 64 1.65 -> 2.75 ms) */
#endif
// 64x64 tiles on the ring body (gt_gemm64.h, round 5): interior problems of GT_T64R_MIN .. GT_T64R_MAX tiles of 64x64 -- too few 128x128
// tiles to fill 256 CUs evenly, enough 64x64 ones (d_model 512 at 2048 tokens: N = 512 -> 256 tiles, QKV -> 768 = 3 per CU)
static inline bool gemm64_ok(const GemmArgs& g, int epi);
template <bool BKM, int EPI>
static inline void gemm64_launch(const GemmArgs& g, hipStream_t s);
static inline bool gemm64h_ok(const GemmArgs& g, int epi);
template <bool BKM, int EPI>
static inline void gemm64h_launch(const GemmArgs& g, hipStream_t s);
#ifndef GT_T64R_MIN
#define GT_T64R_MIN 256         /* one tile per CU at least (192 -- the QKV projection of the d_model-256 YAMLs at batch 32 -- measured 0.648 vs 0.642 ms per step) */
#endif
#ifndef GT_T64R_MAX
#define GT_T64R_MAX 2047        /* 64x64 tiles; from 512 tiles of 128x128 the big tile (half the operand bytes per flop) has two full rounds */
#endif
#ifndef GT_T64H_MAX
#define GT_T64H_MAX 767         /* both operands bf16: 64x64 tiles only below 192 tiles of 128x128 (measured, gemm_bench: M 2048 N 512 6.2 vs 10.4 us, M 8192
                                   N 256 7.7 vs 8.6 -- but M 2048 N 1536 12.9 vs 11.7, M 8192 N 768 18.6 vs 13.7, M 16384 N 512 29 vs 18.6: at the bf16 MFMA rate
                                   the 64x64 tile's doubled operand bytes per flop bind as soon as the big tile has a round of its own) */
#endif
static inline long gt_env_long(const char* name, long dflt) { const char* e = getenv(name); return (e && e[0]) ? atol(e) : dflt; }
static inline bool gemm64_range(long t64, bool h) {
  static const long lo = gt_env_long("GT_T64R_MIN", GT_T64R_MIN), hi = gt_env_long("GT_T64R_MAX", GT_T64R_MAX), hih = gt_env_long("GT_T64H_MAX", GT_T64H_MAX);
  return t64 >= lo && t64 <= (h ? hih : hi);
}
// will gemm_launch put this problem on a kernel with the shared store epilogue of gt_gemm32.h (the only ones that write a bf16 output, C16,
// and can leave the fp32 one out)?  Mirrors the rules below.
template <bool BKM, int EPI>
static inline bool gemm_on_big_kernel(const GemmArgs& g) {
  const long b64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64), b128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
  if (g.bf16 && gemm64_range(b64, true) && gemm64h_ok(g, EPI)) return true;
  if (!(g.bf16 && g.A16 && g.B16) && gemm64_range(b64, false) && gemm64_ok(g, EPI)) return true;
  if (g.bf16 && b128 >= GT_T128H_MIN && gemm32h_ok(g, EPI)) return true;
  return b128 >= GT_T128_BIG_MIN && gemm32_ok(g, EPI, BKM);
}
// standard (non-row) epilogues: pick the tile by how many workgroups the problem yields
template <bool AKM, bool BKM, int EPI>
static inline void gemm_launch(GemmArgs g, hipStream_t s) {
  g.k_chunk = (g.K + 63) / 64 * 64;
  if (EPI == EPI_ATOMIC) {
    const int splitk = wgrad_split(g, 1024);
    if (g.bf16) gemm_launch_cfg<2, 2, 1, 1, 64, AKM, BKM, EPI, 1>(g, splitk, s);
    else        gemm_launch_cfg<2, 2, 1, 1, 64, AKM, BKM, EPI>(g, splitk, s);
    return;
  }
  if constexpr (!AKM && (EPI == EPI_STORE || EPI == EPI_RELU_DROP || EPI == EPI_MASK_NZ || EPI == EPI_ADD_RELUMASK_DROP)) {
    const long b64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
    if (g.bf16 && gemm64_range(b64, true) && gemm64h_ok(g, EPI)) { gemm64h_launch<BKM, EPI>(g, s); return; }       // both operands as bf16 shadows
    if (!(g.bf16 && g.A16 && g.B16) && gemm64_range(b64, false) && gemm64_ok(g, EPI)) { gemm64_launch<BKM, EPI>(g, s); return; }
    const long b128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (g.bf16 && b128 >= GT_T128H_MIN && gemm32h_ok(g, EPI)) { gemm32h_launch<BKM, EPI>(g, s); return; }     // both operands as bf16 shadows
    if (g.bf16 && b128 >= GT_T128_BIG_MIN && gemm32_ok(g, EPI, BKM)) { gemm32_launch<BKM, EPI>(g, s); return; }
  }
  if (g.bf16) {
    // bf16 operands: the same three tile classes; slabs of 64 k (two MFMA k-steps per barrier) except on the 128x128 tile,
    // where 32 keeps two workgroups per CU resident (49 KB of LDS each)
    const long u64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64), u128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (u128 >= GT_T128_MIN) { g.k_chunk = (g.K + 31) / 32 * 32; gemm_launch_cfg<2, 2, 4, 4, 32, AKM, BKM, EPI, 1>(g, 1, s); }
    else if (u64 >= GT_T64_MIN) { g.k_chunk = (g.K + 63) / 64 * 64; gemm_launch_cfg<2, 2, 2, 2, 64, AKM, BKM, EPI, 1>(g, 1, s); }
    else { g.k_chunk = (g.K + 63) / 64 * 64; gemm_launch_cfg<2, 2, 1, 1, 64, AKM, BKM, EPI, 1>(g, 1, s); }
    return;
  }
  const long t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
  const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
  if constexpr (!AKM && (EPI == EPI_STORE || EPI == EPI_RELU_DROP || EPI == EPI_MASK_NZ || EPI == EPI_ADD_RELUMASK_DROP)) {
    if (t128 >= GT_T128_BIG_MIN && gemm32_ok(g, EPI, BKM)) { gemm32_launch<BKM, EPI>(g, s); return; }
  }
  // 128x128 tiles once they still fill the chip twice over: a 64x64x64 slab needs ~38 GB/s of L2->LDS staging per
  // workgroup to keep its MFMAs fed, two resident workgroups ask a CU for more than it delivers (46-70 GB/s measured);
  // 128x128x32 slabs need half of that per flop
  if (t128 >= GT_T128_MIN) { g.k_chunk = (g.K + 31) / 32 * 32; gemm_launch_cfg<2, 2, 4, 4, 32, AKM, BKM, EPI>(g, 1, s); }
  else if (t64 >= GT_T64_MIN) { g.k_chunk = (g.K + GT_T64_BK - 1) / GT_T64_BK * GT_T64_BK; gemm_launch_cfg<2, 2, 2, 2, GT_T64_BK, AKM, BKM, EPI>(g, 1, s); }
  else { g.k_chunk = (g.K + GT_T32_BK - 1) / GT_T32_BK * GT_T32_BK; gemm_launch_cfg<2, 2, 1, 1, GT_T32_BK, AKM, BKM, EPI>(g, 1, s); }
}

// row epilogues: BN = padded d_model.  16-row tiles while the problem is small (M = 2048 gives only 128 of them); 32-row
// tiles (twice the flops per staged weight byte) once they still yield >= 256 workgroups
#ifndef GT_ROW_BM32_MIN
#define GT_ROW_BM32_MIN 8192
#endif
#ifndef GT_ROW_BM64_MIN
#define GT_ROW_BM64_MIN 16384
#endif
// wide rows (d_model > 128) go on to 64-row tiles when even those fill the chip: per flop they stage half the weight bytes
static inline int gemm_row_bm(int M, int N) { return M < GT_ROW_BM32_MIN ? 16 : (N > 128 && M >= GT_ROW_BM64_MIN) ? 64 : 32; }
static inline bool gemm32row_ok(const GemmArgs& g, bool bkm);
template <bool BKM, int EPI>
static inline void gemm32row_launch(const GemmArgs& g, hipStream_t s);
#ifndef GT_ROW32_BM64_MIN_WG
#define GT_ROW32_BM64_MIN_WG 256   /* d_model 256: 64-row tiles from this many workgroups of them, 32-row tiles below */
#endif
#ifndef GT_ROW32_MIN_M
#define GT_ROW32_MIN_M 8192     /* the ring-body row tiles (gt_gemm32.h) from this many tokens: 256 / 128 workgroups of 32 / 64 rows */
#endif
// rows per workgroup of the row-fused launch of g (the LayerNorm-backward partials table has one row per workgroup)
static inline int gemm_row_rows(const GemmArgs& g, bool bkm) {
  if (g.M >= GT_ROW32_MIN_M && gemm32row_ok(g, bkm)) return (g.N == 512 || g.M / 64 >= GT_ROW32_BM64_MIN_WG) ? 64 : 32;
  return gemm_row_bm(g.M, g.N);
}
template <bool AKM, bool BKM, int EPI>
static inline int gemm_launch_row(GemmArgs g, hipStream_t s) {
  if constexpr (!AKM) { if (g.M >= GT_ROW32_MIN_M && gemm32row_ok(g, BKM)) { gemm32row_launch<BKM, EPI>(g, s); return 0; } }
  const int bm = gemm_row_bm(g.M, g.N);
  const bool small = bm == 16;
  if (g.N <= 32) {
    g.k_chunk = (g.K + 63) / 64 * 64;
    if (small) gemm_launch_cfg<1, 2, 1, 1, 64, AKM, BKM, EPI>(g, 1, s);
    else       gemm_launch_cfg<2, 2, 1, 1, 64, AKM, BKM, EPI>(g, 1, s);
  } else if (g.N <= 64) {
    g.k_chunk = (g.K + 63) / 64 * 64;
    if (small) gemm_launch_cfg<1, 4, 1, 1, 64, AKM, BKM, EPI>(g, 1, s);
    else       gemm_launch_cfg<2, 2, 1, 2, 64, AKM, BKM, EPI>(g, 1, s);
  } else if (g.N <= 128) {
    g.k_chunk = (g.K + 63) / 64 * 64;
    g.k_chunk = (g.K + 127) / 128 * 128;
    if (small) gemm_launch_cfg<1, 4, 1, 2, 128, AKM, BKM, EPI>(g, 1, s);
    else       gemm_launch_cfg<2, 2, 1, 4, 64, AKM, BKM, EPI>(g, 1, s);
  } else if (g.N <= 256) {
    g.k_chunk = (g.K + 31) / 32 * 32;
    if (small)        gemm_launch_cfg<1, 4, 1, 4, 32, AKM, BKM, EPI>(g, 1, s);
    else if (bm == 32) gemm_launch_cfg<1, 4, 2, 4, 32, AKM, BKM, EPI>(g, 1, s);
    else              gemm_launch_cfg<1, 4, 4, 4, 32, AKM, BKM, EPI>(g, 1, s);
  } else if (g.N <= 512) {
    g.k_chunk = (g.K + 15) / 16 * 16;
    if (small)        gemm_launch_cfg<1, 4, 1, 8, 16, AKM, BKM, EPI>(g, 1, s);
    else if (bm == 32) gemm_launch_cfg<1, 4, 2, 8, 16, AKM, BKM, EPI>(g, 1, s);
    else              gemm_launch_cfg<1, 4, 4, 8, 16, AKM, BKM, EPI>(g, 1, s);
  } else {
    return -1;
  }
  return 0;
}

#include "gt_gemm32.h"
#include "gt_gemm64.h"
