// Row-chain kernels: everything between two attention kernels of an encoder layer is ROW-LOCAL, so one workgroup
// that owns a 16-row tile runs the whole chain with the intermediates resident in LDS:
//
//   forward   ctx --Wo--> +x, LN1 -> x1 --W1--> relu,drop -> h --W2--> +x1, LN2 -> x2 --(W_qkv of the next layer | final LN)
//   backward  (dqkv_next Win_next + dz1_next | dlogits Wout, LNf') -> LN2' -> dz2 --W2--> *relu' -> dh --W1--> +dz2, LN1' -> dz1
//             --Wo--> dctx
//
// Why: a kernel boundary costs ~4 us on this machine whatever the kernel does, plus a round trip of the activations
// through L2/MALL; at the headline size (M = 2048 rows) the five forward GEMM launches of a layer did ~1.5 us of MFMA
// work each.  The chain replaces 4 launches per layer per direction by 1, never re-reads x1 / h / dz from memory, and
// keeps the weight-slab pipeline (global -> registers -> LDS, double buffered) running ACROSS stage boundaries: the
// first slab of the next stage's weights is already in flight while the current stage's epilogue runs.
// Each GEMM stage is `chain_pass`: A = a 16-row LDS tile, B = weights streamed in BK-wide slabs, 4 waves x TN
// v_mfma_f32_16x16x4_f32 tiles = DPAD output columns per pass.  LayerNorm forward/backward run on 16-lane row groups
// exactly as in gt_gemm.h's row epilogues.
#pragma once
#include "gt_gemm.h"

struct PassDesc { const float* W; int ldw, K, N, n0; };   // W == nullptr: no pass

template <int NP, int BK, bool BKM>
struct ChainB {
  static constexpr int ROWS = BKM ? BK : NP, COLS = BKM ? NP : BK;
  static constexpr int STR = COLS + 4, SZ = ROWS * STR;
  TileStage<ROWS, COLS, 256> st;
  __device__ __forceinline__ void load(const PassDesc& p, int k0, int tid) {
    const bool vec = ((p.ldw & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.W) & 15) == 0);
    if (BKM) st.load(p.W, p.ldw, k0, p.n0, p.K, p.N, vec, tid);     // slab rows = k, cols = n
    else     st.load(p.W, p.ldw, p.n0, k0, p.N, p.K, vec, tid);     // slab rows = n, cols = k
  }
  __device__ __forceinline__ void store(float* sB, int tid) const { st.store(sB, STR, tid); }
};

// acc[t] = sum_k sA[row][k] * B(k, n0 + (wave*TN + t)*16 + l16)   over the whole K of `cur`.
// On return the registers of `cb` hold slab 0 of `nxt` (if any) and `primed` says so.
template <int TN, int BK, bool BKM>
__device__ __forceinline__ void chain_pass(f32x4 (&acc)[TN], const float* sA, int lda, const PassDesc& cur, const PassDesc& nxt,
                                           ChainB<64 * TN, BK, BKM>& cb, bool& primed, float* sB, int tid) {
  typedef ChainB<64 * TN, BK, BKM> CB;
  constexpr int STR = CB::STR, SZ = CB::SZ;
  const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ns = (cur.K + BK - 1) / BK;
  if (!primed) cb.load(cur, 0, tid);
  cb.store(sB, tid);
  __syncthreads();
  primed = false;
  for (int s = 0; s < ns; ++s) {
    const float* b = sB + (s & 1) * SZ;
    if (s + 1 < ns) cb.load(cur, (s + 1) * BK, tid);
    else if (nxt.W != nullptr) { cb.load(nxt, 0, tid); primed = true; }
    // (no tail skip: slabs are zero-filled beyond K, and a branch here splits the MFMA block -- hipcc then moves the
    //  accumulators VGPR<->AGPR around every 8 MFMAs and exposes the LDS latency each time)
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      {
        const float4 a4 = *reinterpret_cast<const float4*>(&sA[l16 * lda + s * BK + kk * 16 + 4 * lg]);
        const float af[4] = {a4.x, a4.y, a4.z, a4.w};
        float bf[TN][4];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int col = (wave * TN + t) * 16 + l16;
          if (!BKM) {
            const float4 t4 = *reinterpret_cast<const float4*>(&b[col * STR + kk * 16 + 4 * lg]);
            bf[t][0] = t4.x; bf[t][1] = t4.y; bf[t][2] = t4.z; bf[t][3] = t4.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[t][j] = b[(kk * 16 + 4 * lg + j) * STR + col];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < TN; ++t) acc[t] = GT_MFMA16(af[j], bf[t][j], acc[t]);
      }
    }
    if (s + 1 < ns) cb.store(sB + ((s + 1) & 1) * SZ, tid);
    __syncthreads();
  }
}

template <int DPAD>
struct ChainCfg {
  static constexpr int TN = DPAD / 64, NP = DPAD, BK = DPAD <= 128 ? 64 : 32;
  static constexpr int XSTR = DPAD + 4;
  static constexpr int FMAX = 512;                                   // dim_feedforward the hidden LDS tile is sized for
  static constexpr int HW = (3 * DPAD > FMAX ? 3 * DPAD : FMAX);     // backward also parks the (16, 3d) dqkv tile there
  static constexpr int HSTR = HW + 4;
};
static inline bool chain_supported(int d, int F) { return (d % 16) == 0 && d <= 256 && F <= 512; }
static inline int chain_dpad(int d) { return d <= 64 ? 64 : d <= 128 ? 128 : 256; }

__device__ __forceinline__ float chain_ldg(const float* p, size_t idx, bool ok) { return *(ok ? p + idx : gt_zero_page); }

// stage the (rows row0.., cols 0..K) block of a global (M, ld) matrix into an LDS tile [16][str], zero beyond M / K
// up to the next multiple of 16 columns (the MFMA k-chunks read whole 16-wide groups)
__device__ __forceinline__ void chain_load_tile(float* s, int str, const float* src, int ld, int row0, int M, int K, int tid) {
  const int K16 = (K + 15) / 16 * 16;
  const bool vec = ((ld & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  if (vec) {
    for (int ch = tid; ch < 16 * (K16 / 4); ch += 256) {
      const int r = ch / (K16 / 4), c = (ch % (K16 / 4)) * 4;
      const bool ok = row0 + r < M && c < K;
      *reinterpret_cast<float4*>(&s[r * str + c]) = *reinterpret_cast<const float4*>(ok ? src + (size_t)(row0 + r) * ld + c : gt_zero_page);
    }
  } else {
    for (int e = tid; e < 16 * K16; e += 256) {
      const int r = e / K16, c = e % K16;
      s[r * str + c] = chain_ldg(src, (size_t)(row0 + r) * ld + c, row0 + r < M && c < K);
    }
  }
}

// dgamma/dbeta partials of one LayerNorm backward inside a chain kernel: 16 row groups -> LDS -> part[tile][2][N]
template <int DPAD>
__device__ __forceinline__ void chain_ln_partials(const float (&dg)[DPAD / 16], const float (&db)[DPAD / 16], float* sRed, float* part,
                                                  int N, int tid) {
  constexpr int XSTR = DPAD + 4;
  const int lane = tid & 63, l16 = lane & 15, grp = (tid >> 6) * 4 + (lane >> 4);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < DPAD / 16; ++i) {
    sRed[grp * XSTR + l16 + 16 * i] = dg[i];
    sRed[(16 + grp) * XSTR + l16 + 16 * i] = db[i];
  }
  __syncthreads();
  for (int c = tid; c < N; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { a += sRed[q * XSTR + c]; b += sRed[(16 + q) * XSTR + c]; }
    part[((size_t)blockIdx.x * 2) * N + c] = a;
    part[((size_t)blockIdx.x * 2 + 1) * N + c] = b;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ forward chain
struct ChainFwdArgs {
  int M, d, F;
  const float* ctx; const float* xin;
  const float *Wo, *bo, *g1, *be1, *W1, *b1, *W2, *b2, *g2, *be2;
  float *x1, *xhat1, *rstd1, *hact, *xout, *xhat2, *rstd2;
  const float *gf, *bef; float *fin, *xhatf, *rstdf;        // final norm (last layer) or gf == nullptr
  const float *Wqkv, *bqkv; float* qkv;                      // next layer's packed in-projection or Wqkv == nullptr
  DropArgs drop1, dropH, dropF;
};

template <int DPAD>
__global__ __launch_bounds__(256) void chain_fwd_kernel(ChainFwdArgs a) {
  typedef ChainCfg<DPAD> C;
  constexpr int TN = C::TN, NP = C::NP, BK = C::BK, XSTR = C::XSTR, HSTR = C::HSTR, CPL = DPAD / 16;
  typedef ChainB<NP, BK, false> CB;
  __shared__ __attribute__((aligned(16))) float smem[2 * 16 * XSTR + 16 * HSTR + 2 * CB::SZ];
  float* sX = smem;                       // x1, then x2: the A operand of FFN1 / next QKV
  float* sC = sX + 16 * XSTR;             // accumulator staging for the LayerNorm row pass
  float* sH = sC + 16 * XSTR;             // ctx tile first, then the hidden activation (16, F)
  float* sB = sH + 16 * HSTR;             // weight slabs, double buffered
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const int row0 = blockIdx.x * 16, d = a.d, F = a.F;
  const int grp = wave * 4 + lg, grow = row0 + grp;         // the row this 16-lane group owns in row passes
  const bool live = grow < a.M;
  const size_t growc = live ? grow : 0;
  const float invN = 1.0f / (float)d;
  CB cb;
  bool primed = false;
  f32x4 acc[TN];

  // LDS starts as garbage; A tiles are read in whole BK-wide slabs (zero weights beyond K), and 0 * NaN = NaN
  for (int e = tid; e < 2 * 16 * XSTR + 16 * HSTR; e += 256) smem[e] = 0.f;
  __syncthreads();
  chain_load_tile(sH, XSTR, a.ctx, d, row0, a.M, d, tid);

  // ---- stage 1: out-proj + bias, dropout1, + x, LayerNorm1 -> x1
  {
    const PassDesc cur{a.Wo, d, d, d, 0}, nxt{a.W1, d, d, F, 0};
    chain_pass<TN, BK, false>(acc, sH, XSTR, cur, nxt, cb, primed, sB, tid);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    __syncthreads();
    const uint32_t key = gt_drop_key(a.drop1);
    float z[CPL], e1[CPL], e2[CPL], e3[CPL], e4[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      const bool okc = c < d;
      z[i] = sC[grp * XSTR + c];
      e1[i] = chain_ldg(a.bo, c, okc); e2[i] = chain_ldg(a.xin, growc * d + c, okc);
      e3[i] = chain_ldg(a.g1, c, okc); e4[i] = chain_ldg(a.be1, c, okc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      z[i] = (c < d) ? (z[i] + e1[i]) * gt_drop_mul(a.drop1, key, (uint32_t)(growc * d + c)) + e2[i] : 0.f;
      s += z[i];
    }
    const float mean = gt_red16(s) * invN;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { if (l16 + 16 * i < d) { const float dd = z[i] - mean; q += dd * dd; } }
    const float rstd = 1.0f / sqrtf(gt_red16(q) * invN + GT_LN_EPS);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      if (c < d) {
        const float xh = (z[i] - mean) * rstd, y = xh * e3[i] + e4[i];
        sX[grp * XSTR + c] = y;
        if (live) { a.xhat1[(size_t)grow * d + c] = xh; a.x1[(size_t)grow * d + c] = y; }
      }
    }
    if (live && l16 == 0) a.rstd1[grow] = rstd;
  }

  // ---- stage 2: h = dropout(relu(x1 W1^T + b1)), DPAD columns per pass; h stays in LDS and goes to memory for backward
  {
    const uint32_t key = gt_drop_key(a.dropH);
    const int np = (F + NP - 1) / NP, F16 = (F + 15) / 16 * 16;
    for (int p = 0; p < np; ++p) {
      const PassDesc cur{a.W1, d, d, F, p * NP};
      const PassDesc nxt = (p + 1 < np) ? PassDesc{a.W1, d, d, F, (p + 1) * NP} : PassDesc{a.W2, F, F, d, 0};
      chain_pass<TN, BK, false>(acc, sX, XSTR, cur, nxt, cb, primed, sB, tid);
      float bia[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t) { const int col = p * NP + (wave * TN + t) * 16 + l16; bia[t] = chain_ldg(a.b1, col, col < F); }
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 4 * lg + r, col = p * NP + (wave * TN + t) * 16 + l16, row = row0 + rl;
          if (col < F) {
            const float v = fmaxf(acc[t][r] + bia[t], 0.f) * gt_drop_mul(a.dropH, key, (uint32_t)((size_t)row * F + col));
            sH[rl * HSTR + col] = v;
            if (row < a.M) a.hact[(size_t)row * F + col] = v;
          } else if (col < F16) {
            sH[rl * HSTR + col] = 0.f;
          }
        }
    }
  }

  // ---- stage 3: FFN linear2 + bias, dropout, + x1, LayerNorm2 -> x2 (and the final encoder norm on the last layer)
  {
    const PassDesc cur{a.W2, F, F, d, 0};
    const PassDesc nxt = a.Wqkv ? PassDesc{a.Wqkv, d, d, 3 * d, 0} : PassDesc{nullptr, 0, 0, 0, 0};
    chain_pass<TN, BK, false>(acc, sH, HSTR, cur, nxt, cb, primed, sB, tid);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    __syncthreads();
    const uint32_t key = gt_drop_key(a.dropF);
    float z[CPL], e1[CPL], e3[CPL], e4[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      const bool okc = c < d;
      z[i] = sC[grp * XSTR + c];
      e1[i] = chain_ldg(a.b2, c, okc); e3[i] = chain_ldg(a.g2, c, okc); e4[i] = chain_ldg(a.be2, c, okc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      z[i] = (c < d) ? (z[i] + e1[i]) * gt_drop_mul(a.dropF, key, (uint32_t)(growc * d + c)) + sX[grp * XSTR + c] : 0.f;
      s += z[i];
    }
    const float mean = gt_red16(s) * invN;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { if (l16 + 16 * i < d) { const float dd = z[i] - mean; q += dd * dd; } }
    const float rstd = 1.0f / sqrtf(gt_red16(q) * invN + GT_LN_EPS);
    float y[CPL];
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      y[i] = 0.f;
      if (c < d) {
        const float xh = (z[i] - mean) * rstd;
        y[i] = xh * e3[i] + e4[i];
        sX[grp * XSTR + c] = y[i];
        if (live) { a.xhat2[(size_t)grow * d + c] = xh; a.xout[(size_t)grow * d + c] = y[i]; }
      }
      s2 += y[i];
    }
    if (live && l16 == 0) a.rstd2[grow] = rstd;
    if (a.gf != nullptr) {               // Encoder.Encoder.norm on top of the last layer's output
      const float mean2 = gt_red16(s2) * invN;
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) { if (l16 + 16 * i < d) { const float dd = y[i] - mean2; q2 += dd * dd; } }
      const float rstdf = 1.0f / sqrtf(gt_red16(q2) * invN + GT_LN_EPS);
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        if (c < d && live) {
          const float xh = (y[i] - mean2) * rstdf;
          a.xhatf[(size_t)grow * d + c] = xh;
          a.fin[(size_t)grow * d + c] = xh * chain_ldg(a.gf, c, true) + chain_ldg(a.bef, c, true);
        }
      }
      if (live && l16 == 0) a.rstdf[grow] = rstdf;
    }
  }

  // ---- stage 4: the NEXT layer's packed q,k,v in-projection of x2
  if (a.Wqkv != nullptr) {
    const int N3 = 3 * d, np = (N3 + NP - 1) / NP;
    for (int p = 0; p < np; ++p) {
      const PassDesc cur{a.Wqkv, d, d, N3, p * NP};
      const PassDesc nxt = (p + 1 < np) ? PassDesc{a.Wqkv, d, d, N3, (p + 1) * NP} : PassDesc{nullptr, 0, 0, 0, 0};
      chain_pass<TN, BK, false>(acc, sX, XSTR, cur, nxt, cb, primed, sB, tid);
      float bia[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t) { const int col = p * NP + (wave * TN + t) * 16 + l16; bia[t] = chain_ldg(a.bqkv, col, col < N3); }
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = p * NP + (wave * TN + t) * 16 + l16, row = row0 + 4 * lg + r;
          if (col < N3 && row < a.M) a.qkv[(size_t)row * N3 + col] = acc[t][r] + bia[t];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward chain
struct ChainBwdArgs {
  int M, d, F;
  const float* A0; int K0;              // top layer: dlogits (M,27), K0 = 27; else dqkv of the layer above (M,3d), K0 = 3d
  const float* W0;                      // top: OutputLayer weight (27,d); else the layer above's in_proj_weight (3d,d)  [k][n]
  const float* res0;                    // dz1 of the layer above (residual path) or nullptr (top)
  const float *xhatf, *rstdf, *gf; float* partf;            // final norm backward (top layer only; gf == nullptr otherwise)
  const float *xhat2, *rstd2, *g2; float* part2; DropArgs dropF;
  float* dz2m_out;                      // (M,d) masked grad of linear2's output: wgrad input, kept per layer
  const float *W2, *hact; float mask_scale; float* dhid_out;
  const float* W1;
  const float *xhat1, *rstd1, *g1; float* part1; DropArgs drop1;
  float *dz1_out, *dz1m_out;
  const float* Wo; float* dctx_out;
};

// LayerNorm backward of one row held by a 16-lane group: dy[] in, dz[] out; accumulates dg/db
template <int CPL>
__device__ __forceinline__ void chain_ln_bwd_row(float (&dy)[CPL], const float* xhat, const float* rstd, const float* gamma, size_t growc,
                                                 bool live, int d, int l16, float invN, float (&dg)[CPL], float (&db)[CPL]) {
  float xh[CPL], ga[CPL];
  const float rs = rstd[growc];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = l16 + 16 * i;
    const bool ok = live && c < d;
    xh[i] = chain_ldg(xhat, growc * d + c, ok);
    ga[i] = chain_ldg(gamma, c, ok);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const float gdy = dy[i] * ga[i];
    s1 += gdy; s2 += gdy * xh[i];
    dg[i] += dy[i] * xh[i]; db[i] += dy[i];
  }
  const float m1 = gt_red16(s1) * invN, m2 = gt_red16(s2) * invN;
#pragma unroll
  for (int i = 0; i < CPL; ++i) dy[i] = (live && l16 + 16 * i < d) ? rs * (dy[i] * ga[i] - m1 - xh[i] * m2) : 0.f;
}

template <int DPAD>
__global__ __launch_bounds__(256) void chain_bwd_kernel(ChainBwdArgs a) {
  typedef ChainCfg<DPAD> C;
  constexpr int TN = C::TN, NP = C::NP, BK = (DPAD <= 128 ? 64 : 16), XSTR = C::XSTR, HSTR = C::HSTR, CPL = DPAD / 16;
  typedef ChainB<NP, BK, true> CB;
  __shared__ __attribute__((aligned(16))) float smem[2 * 16 * XSTR + 32 * XSTR + 16 * HSTR + 2 * CB::SZ];
  float* sX = smem;                       // dz2m, then dz1m: A operand of the W2 / Wo passes
  float* sR = sX + 16 * XSTR;             // dz2 (unmasked): residual into LayerNorm1's backward
  float* sC = sR + 16 * XSTR;             // staging / dgamma-dbeta reduction scratch (32 rows)
  float* sH = sC + 32 * XSTR;             // A0 tile first (dlogits or dqkv of the layer above), then dhid (16, F)
  float* sB = sH + 16 * HSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const int row0 = blockIdx.x * 16, d = a.d, F = a.F;
  const int grp = wave * 4 + lg, grow = row0 + grp;
  const bool live = grow < a.M;
  const size_t growc = live ? grow : 0;
  const float invN = 1.0f / (float)d;
  CB cb;
  bool primed = false;
  f32x4 acc[TN];
  float dg[CPL], db[CPL];

  for (int e = tid; e < 2 * 16 * XSTR + 32 * XSTR + 16 * HSTR; e += 256) smem[e] = 0.f;     // see chain_fwd_kernel
  __syncthreads();
  chain_load_tile(sH, HSTR, a.A0, a.K0, row0, a.M, a.K0, tid);

  // ---- stage 0: grad w.r.t. this layer's output, then (final norm's and) LayerNorm2's backward -> dz2, dz2m
  {
    const PassDesc cur{a.W0, d, a.K0, d, 0}, nxt{a.W2, F, d, F, 0};
    chain_pass<TN, BK, true>(acc, sH, HSTR, cur, nxt, cb, primed, sB, tid);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    __syncthreads();
    float dy[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      const bool ok = live && c < d;
      dy[i] = ok ? sC[grp * XSTR + c] + chain_ldg(a.res0, growc * d + c, ok && a.res0 != nullptr) : 0.f;
    }
    if (a.gf != nullptr) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) { dg[i] = 0.f; db[i] = 0.f; }
      chain_ln_bwd_row<CPL>(dy, a.xhatf, a.rstdf, a.gf, growc, live, d, l16, invN, dg, db);
      chain_ln_partials<DPAD>(dg, db, sC, a.partf, d, tid);
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    chain_ln_bwd_row<CPL>(dy, a.xhat2, a.rstd2, a.g2, growc, live, d, l16, invN, dg, db);
    const uint32_t key = gt_drop_key(a.dropF);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      if (c < d) {
        const float m = dy[i] * gt_drop_mul(a.dropF, key, (uint32_t)(growc * d + c));
        sR[grp * XSTR + c] = dy[i];
        sX[grp * XSTR + c] = m;
        if (live) a.dz2m_out[(size_t)grow * d + c] = m;
      }
    }
    chain_ln_partials<DPAD>(dg, db, sC, a.part2, d, tid);
  }

  // ---- stage 1: dhid = (dz2m W2) * relu'/dropout mask (from the saved activation), kept in LDS and stored for wgrad
  {
    const int np = (F + NP - 1) / NP, F16 = (F + 15) / 16 * 16;
    for (int p = 0; p < np; ++p) {
      const PassDesc cur{a.W2, F, d, F, p * NP};
      const PassDesc nxt = (p + 1 < np) ? PassDesc{a.W2, F, d, F, (p + 1) * NP} : PassDesc{a.W1, d, F, d, 0};
      chain_pass<TN, BK, true>(acc, sX, XSTR, cur, nxt, cb, primed, sB, tid);
      float hv[TN][4];
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = p * NP + (wave * TN + t) * 16 + l16, row = row0 + 4 * lg + r;
          hv[t][r] = chain_ldg(a.hact, (size_t)row * F + col, col < F && row < a.M);
        }
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 4 * lg + r, col = p * NP + (wave * TN + t) * 16 + l16, row = row0 + rl;
          if (col < F) {
            const float v = (hv[t][r] != 0.f) ? acc[t][r] * a.mask_scale : 0.f;
            sH[rl * HSTR + col] = v;
            if (row < a.M) a.dhid_out[(size_t)row * F + col] = v;
          } else if (col < F16) {
            sH[rl * HSTR + col] = 0.f;
          }
        }
    }
  }

  // ---- stage 2: dx1 = dhid W1 + dz2, LayerNorm1's backward -> dz1 (residual for the layer below), dz1m
  {
    const PassDesc cur{a.W1, d, F, d, 0}, nxt{a.Wo, d, d, d, 0};
    chain_pass<TN, BK, true>(acc, sH, HSTR, cur, nxt, cb, primed, sB, tid);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    __syncthreads();
    float dy[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      dy[i] = (live && c < d) ? sC[grp * XSTR + c] + sR[grp * XSTR + c] : 0.f;
      dg[i] = 0.f; db[i] = 0.f;
    }
    chain_ln_bwd_row<CPL>(dy, a.xhat1, a.rstd1, a.g1, growc, live, d, l16, invN, dg, db);
    const uint32_t key = gt_drop_key(a.drop1);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      if (c < d) {
        const float m = dy[i] * gt_drop_mul(a.drop1, key, (uint32_t)(growc * d + c));
        sX[grp * XSTR + c] = m;
        if (live) { a.dz1_out[(size_t)grow * d + c] = dy[i]; a.dz1m_out[(size_t)grow * d + c] = m; }
      }
    }
    chain_ln_partials<DPAD>(dg, db, sC, a.part1, d, tid);
  }

  // ---- stage 3: dctx = dz1m Wo (the attention backward kernel takes it from memory)
  {
    const PassDesc cur{a.Wo, d, d, d, 0}, nxt{nullptr, 0, 0, 0, 0};
    chain_pass<TN, BK, true>(acc, sX, XSTR, cur, nxt, cb, primed, sB, tid);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = (wave * TN + t) * 16 + l16, row = row0 + 4 * lg + r;
        if (col < d && row < a.M) a.dctx_out[(size_t)row * d + col] = acc[t][r];
      }
  }
}
