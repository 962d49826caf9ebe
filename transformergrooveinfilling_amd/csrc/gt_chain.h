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

// ---- weight-slab stream: loader waves -> LDS double buffer -> compute waves ------------------------------------------
// WAVE SPECIALISATION.  A workgroup is 8 waves, two per SIMD.  Waves 4-7 ("loaders") stream the weights of ALL passes of
// the kernel, slab by slab (NP output columns x BK of k), global -> their own registers -> LDS, with TWO slabs in flight
// in registers; waves 0-3 read LDS and issue MFMA and run the epilogues.  One workgroup barrier per slab hands slab g to
// the compute waves and LDS slot (g+1)&1 back to the loaders.
// How this shape was arrived at (profiles/, tools/ubench/):
//   * one set of waves doing both jobs was INSTRUCTION-ISSUE bound: ~240 instructions per slab in one wave per SIMD,
//     address generation and MFMAs back to back -- 45-50 us per kernel, still 35-41 us with the MFMAs or the loads
//     compiled out;
//   * LDS-DMA (global_load_lds) with 3 slabs in flight was no faster: its fill cadence here is ~16 KB per ~1400 cycles per
//     CU (~27 GB/s), while plain loads through registers stream an L2-resident block at 46-70 GB/s per workgroup.
// Loaders never run epilogue code; they only take the same barriers (`nbar_at`), so both paths execute identical barrier
// sequences (the host emulator asserts that).
template <int NP, int BK, bool BKM>
struct SlabRegs {
  static constexpr int ROWS = BKM ? BK : NP, COLS = BKM ? NP : BK;
  static constexpr int STR = COLS + 4, SZ = ROWS * STR;
  typedef TileStage<ROWS, COLS, 256> TS;
  TS st;
  // Fast path: when every element of a pass's slabs is in range, a lane's source address is (wave-uniform slab base) +
  // (a per-lane 32-bit offset that is constant for the whole pass), so issuing a chunk is ONE load instruction.  The
  // generic TileStage path costs ~20 VALU ops per chunk (div/mod, bounds, 64-bit multiply) -- measured 1300 cycles per
  // 32 KB slab in the loader waves, more than the 32 MFMAs the compute waves spend on it.
  int off[TS::PER];
  int cached_pass;
  __device__ __forceinline__ void init(const float* zp) { st.zp = zp; cached_pass = -1; }
  __device__ __forceinline__ void load(const PassDesc& p, int pass_id, int k0, int ltid) {
    const bool vec = ((p.ldw & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.W) & 15) == 0);
    const bool fast = vec && (p.n0 + NP <= p.N) && (k0 + BK <= p.K) && (TS::CH % 256 == 0);
    if (fast) {
      if (cached_pass != pass_id) {
        cached_pass = pass_id;
#pragma unroll
        for (int i = 0; i < TS::PER; ++i) {
          const int ch = ltid + i * 256, r = ch / TS::CPR, c = (ch % TS::CPR) * 4;
          off[i] = r * p.ldw + c;
        }
      }
      const float* base = BKM ? p.W + ((size_t)k0 * p.ldw + p.n0) : p.W + ((size_t)p.n0 * p.ldw + k0);
#pragma unroll
      for (int i = 0; i < TS::PER; ++i) st.v[i] = *reinterpret_cast<const float4*>(base + off[i]);
      return;
    }
    if (BKM) st.load(p.W, p.ldw, k0, p.n0, p.K, p.N, vec, ltid);     // slab rows = k, cols = n
    else     st.load(p.W, p.ldw, p.n0, k0, p.N, p.K, vec, ltid);     // slab rows = n, cols = k
  }
  __device__ __forceinline__ void store(float* slot, int ltid) const { st.store(slot, STR, ltid); }
};
struct SlabCursor { int pi, si; };
// epilogue input, issued BEFORE the pass it belongs to (branch-free: out-of-range lanes read the zero page)
__device__ __forceinline__ void chain_gload(float& dst, const float* p, size_t idx, bool ok, const float* zp) { dst = *(ok ? p + idx : zp); }

template <int TN, int BK, bool BKM, typename DescFn, typename NbarFn>
__device__ __forceinline__ void chain_loader(float* ring, int npass, const DescFn& desc_at, const NbarFn& nbar_at, int ltid,
                                             const float* zp) {
  typedef SlabRegs<64 * TN, BK, BKM> SR;
  SR r0, r1;
  r0.init(zp); r1.init(zp);
  SlabCursor lc{0, 0}, sc{0, 0};
  auto load_next = [&](SR& r) {
    if (lc.pi < npass) {
      const PassDesc dsc = desc_at(lc.pi);
      r.load(dsc, lc.pi, lc.si * BK, ltid);
      if (++lc.si >= (dsc.K + BK - 1) / BK) { lc.si = 0; ++lc.pi; }
    }
  };
  int g = 0;
  // publish slab g, refill its registers with slab g+2, take the barriers the compute waves take for slab g
  auto step = [&](SR& r) {
    r.store(ring + (g & 1) * SR::SZ, ltid);
    ++g;
    load_next(r);
    GT_BARRIER();
    const PassDesc dsc = desc_at(sc.pi);
    if (++sc.si >= (dsc.K + BK - 1) / BK) {
      const int nb = nbar_at(sc.pi);
      for (int i = 0; i < nb; ++i) GT_BARRIER();
      sc.si = 0; ++sc.pi;
    }
  };
  load_next(r0);
  load_next(r1);
  GT_BARRIER();                                   // the compute waves' prologue barrier
  while (sc.pi < npass) {
    step(r0);
    if (sc.pi >= npass) break;
    step(r1);
  }
}

// compute waves: acc[t] = sum_k sA[row][k] * B(k, n0 + (wave*TN + t)*16 + l16) over the whole K of pass `cur`
template <int TN, int BK, bool BKM>
__device__ __forceinline__ void chain_pass(f32x4 (&acc)[TN], const float* sA, int lda, int K, const float* ring, int& gslab, int tid) {
  typedef SlabRegs<64 * TN, BK, BKM> SR;
  constexpr int STR = SR::STR;
  const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ns = (K + BK - 1) / BK;
  for (int s = 0; s < ns; ++s) {
    GT_BARRIER();
    const float* b = ring + (gslab & 1) * SR::SZ;
    ++gslab;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
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
}

template <int DPAD>
struct ChainCfg {
  static constexpr int TN = DPAD / 64, NP = DPAD;
  static constexpr int BK_FWD = DPAD <= 128 ? 64 : 32;               // slab = DPAD columns x BK of k, two LDS slots
  static constexpr int BK_BWD = DPAD <= 128 ? 64 : 16;
  static constexpr int XSTR = DPAD + 4;
  static constexpr int FMAX = 512;                                   // dim_feedforward the hidden LDS tile is sized for
  static constexpr int HW = (3 * DPAD > FMAX ? 3 * DPAD : FMAX);     // backward also parks the (16, 3d) dqkv tile there
  static constexpr int HSTR = HW + 4;
};
static inline bool chain_supported(int d, int F) { return (d % 16) == 0 && d <= 256 && F <= 512 && (F % 4) == 0; }
static inline int chain_dpad(int d) { return d <= 64 ? 64 : d <= 128 ? 128 : 256; }



// stage the (rows row0.., cols 0..K) block of a global (M, ld) matrix into an LDS tile [16][str], zero beyond M / K
// up to the next multiple of 16 columns (the MFMA k-chunks read whole 16-wide groups)
__device__ __forceinline__ void chain_load_tile(float* s, int str, const float* src, int ld, int row0, int M, int K, int tid, const float* zp) {
  const int K16 = (K + 15) / 16 * 16;
  const bool vec = ((ld & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  if (vec) {
    for (int ch = tid; ch < 16 * (K16 / 4); ch += 256) {
      const int r = ch / (K16 / 4), c = (ch % (K16 / 4)) * 4;
      const bool ok = row0 + r < M && c < K;
      *reinterpret_cast<float4*>(&s[r * str + c]) = *reinterpret_cast<const float4*>(ok ? src + (size_t)(row0 + r) * ld + c : zp);
    }
  } else {
    for (int e = tid; e < 16 * K16; e += 256) {
      const int r = e / K16, c = e % K16;
      s[r * str + c] = *((row0 + r < M && c < K) ? src + (size_t)(row0 + r) * ld + c : zp);
    }
  }
}

// dgamma/dbeta partials of one LayerNorm backward inside a chain kernel: 16 row groups -> LDS -> part[tile][2][N]
template <int DPAD>
__device__ __forceinline__ void chain_ln_partials(const float (&dg)[DPAD / 16], const float (&db)[DPAD / 16], float* sRed, float* part,
                                                  int N, int tid, bool cw) {
  constexpr int XSTR = DPAD + 4;
  const int lane = tid & 63, l16 = lane & 15, grp = ((tid >> 6) & 3) * 4 + (lane >> 4);
  GT_BARRIER();                            // every wave of the workgroup takes the barriers; only compute waves work
  if (cw) {
#pragma unroll
    for (int i = 0; i < DPAD / 16; ++i) {
      sRed[grp * XSTR + l16 + 16 * i] = dg[i];
      sRed[(16 + grp) * XSTR + l16 + 16 * i] = db[i];
    }
  }
  GT_BARRIER();
  if (cw) {
    for (int c = tid; c < N; c += 256) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) { a += sRed[q * XSTR + c]; b += sRed[(16 + q) * XSTR + c]; }
      part[((size_t)blockIdx.x * 2) * N + c] = a;
      part[((size_t)blockIdx.x * 2 + 1) * N + c] = b;
    }
  }
  GT_BARRIER();
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
__global__ __launch_bounds__(512) void chain_fwd_kernel(ChainFwdArgs a) {
  typedef ChainCfg<DPAD> C;
  constexpr int TN = C::TN, NP = C::NP, BK = C::BK_FWD, XSTR = C::XSTR, HSTR = C::HSTR, CPL = DPAD / 16;
  typedef SlabRegs<NP, BK, false> SR;
  __shared__ __attribute__((aligned(16))) float work[2 * 16 * XSTR + 16 * HSTR];
  __shared__ __attribute__((aligned(16))) float ring[2 * SR::SZ];
  float* sX = work;                       // x1, then x2: the A operand of FFN1 / next QKV
  float* sC = sX + 16 * XSTR;             // accumulator staging for the LayerNorm row pass
  float* sH = sC + 16 * XSTR;             // ctx tile first, then the hidden activation (16, F)
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, l16 = lane & 15, lg = lane >> 4;
  const bool cw = tid < 256;                                // waves 0-3 compute (MFMA, epilogues); waves 4-7 stream weights
  const int row0 = blockIdx.x * 16, d = a.d, F = a.F;
  const int grp = wave * 4 + lg, grow = row0 + grp;         // the row this 16-lane group owns in row passes
  const bool live = grow < a.M;
  const size_t growc = live ? grow : 0;
  const float invN = 1.0f / (float)d;
  f32x4 acc[TN];
  // the kernel's pass list (weights streamed in this order): Wo | W1 in DPAD-column passes | W2 | next layer's Wqkv passes
  const int np2 = (F + NP - 1) / NP, np4 = a.Wqkv ? (3 * d + NP - 1) / NP : 0, npass = 2 + np2 + np4;
  auto desc_at = [&](int pi) -> PassDesc {
    if (pi == 0) return PassDesc{a.Wo, d, d, d, 0};
    if (pi <= np2) return PassDesc{a.W1, d, d, F, (pi - 1) * NP};
    if (pi == np2 + 1) return PassDesc{a.W2, F, F, d, 0};
    return PassDesc{a.Wqkv, d, d, 3 * d, (pi - np2 - 2) * NP};
  };
  // LDS starts as garbage; A tiles are read in whole BK-wide slabs (zero weights beyond K), and 0 * NaN = NaN
  const float* const zp = gt_zero_ptr();
  if (!cw) {                               // loader waves: stream every pass's weights, mirror the barrier sequence, done
    auto nbar_at = [&](int pi) -> int { return (pi == 0 || pi == np2 + 1) ? 1 : 0; };      // the two LayerNorm epilogues
    chain_loader<TN, BK, false>(ring, npass, desc_at, nbar_at, tid - 256, zp);
    return;
  }
  int gslab = 0;
  for (int e = tid; e < 2 * 16 * XSTR + 16 * HSTR; e += 256) work[e] = 0.f;
  GT_BARRIER();
  if (cw) chain_load_tile(sH, XSTR, a.ctx, d, row0, a.M, d, tid, zp);

  // ---- stage 1: out-proj + bias, dropout1, + x, LayerNorm1 -> x1
  {
    float e1[CPL], e2[CPL], e3[CPL], e4[CPL];                 // epilogue inputs: in flight during the pass
    if (cw) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        const bool okc = c < d;
        chain_gload(e1[i], a.bo, c, okc, zp); chain_gload(e2[i], a.xin, growc * d + c, okc, zp);
        chain_gload(e3[i], a.g1, c, okc, zp); chain_gload(e4[i], a.be1, c, okc, zp);
      }
    }
    chain_pass<TN, BK, false>(acc, sH, XSTR, desc_at(0).K, ring, gslab, tid);
    if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    }
    GT_BARRIER();
    if (cw) {
    const uint32_t key = gt_drop_key(a.drop1);
    float z[CPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      z[i] = (c < d) ? (sC[grp * XSTR + c] + e1[i]) * gt_drop_mul(a.drop1, key, (uint32_t)(growc * d + c)) + e2[i] : 0.f;
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
  }
  // ---- stage 2: h = dropout(relu(x1 W1^T + b1)), DPAD columns per pass; h stays in LDS and goes to memory for backward
  {
    const uint32_t key = gt_drop_key(a.dropH);
    const int F16 = (F + 15) / 16 * 16;
    for (int p = 0; p < np2; ++p) {
      float bia[TN];
      if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t) { const int col = p * NP + (wave * TN + t) * 16 + l16; chain_gload(bia[t], a.b1, col, col < F, zp); }
      }
      chain_pass<TN, BK, false>(acc, sX, XSTR, desc_at(1 + p).K, ring, gslab, tid);
      if (cw) {
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
  }
  // ---- stage 3: FFN linear2 + bias, dropout, + x1, LayerNorm2 -> x2 (and the final encoder norm on the last layer)
  {
    float e1[CPL], e3[CPL], e4[CPL], e5[CPL], e6[CPL];
    if (cw) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        const bool okc = c < d;
        chain_gload(e1[i], a.b2, c, okc, zp); chain_gload(e3[i], a.g2, c, okc, zp); chain_gload(e4[i], a.be2, c, okc, zp);
        chain_gload(e5[i], a.gf, c, okc && a.gf != nullptr, zp); chain_gload(e6[i], a.bef, c, okc && a.gf != nullptr, zp);
      }
    }
    chain_pass<TN, BK, false>(acc, sH, HSTR, desc_at(np2 + 1).K, ring, gslab, tid);
    if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    }
    GT_BARRIER();
    if (cw) {
    const uint32_t key = gt_drop_key(a.dropF);
    float z[CPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      z[i] = (c < d) ? (sC[grp * XSTR + c] + e1[i]) * gt_drop_mul(a.dropF, key, (uint32_t)(growc * d + c)) + sX[grp * XSTR + c] : 0.f;
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
          a.fin[(size_t)grow * d + c] = xh * e5[i] + e6[i];
        }
      }
      if (live && l16 == 0) a.rstdf[grow] = rstdf;
    }
    }
  }
  // ---- stage 4: the NEXT layer's packed q,k,v in-projection of x2
  if (a.Wqkv != nullptr) {
    const int N3 = 3 * d;
    for (int p = 0; p < np4; ++p) {
      float bia[TN];
      if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t) { const int col = p * NP + (wave * TN + t) * 16 + l16; chain_gload(bia[t], a.bqkv, col, col < N3, zp); }
      }
      chain_pass<TN, BK, false>(acc, sX, XSTR, desc_at(np2 + 2 + p).K, ring, gslab, tid);
      if (cw) {
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

// LayerNorm backward of one row held by a 16-lane group: dy[] in, dz[] out (xh / ga / rs preloaded); accumulates dg/db
template <int CPL>
__device__ __forceinline__ void chain_ln_bwd_row(float (&dy)[CPL], const float (&xh)[CPL], const float (&ga)[CPL], float rs, bool live, int d,
                                                 int l16, float invN, float (&dg)[CPL], float (&db)[CPL]) {
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
__global__ __launch_bounds__(512) void chain_bwd_kernel(ChainBwdArgs a) {
  typedef ChainCfg<DPAD> C;
  constexpr int TN = C::TN, NP = C::NP, BK = C::BK_BWD, XSTR = C::XSTR, HSTR = C::HSTR, CPL = DPAD / 16;
  typedef SlabRegs<NP, BK, true> SR;
  __shared__ __attribute__((aligned(16))) float work[2 * 16 * XSTR + 32 * XSTR + 16 * HSTR];    // see chain_fwd_kernel
  __shared__ __attribute__((aligned(16))) float ring[2 * SR::SZ];
  float* sX = work;                       // dz2m, then dz1m: A operand of the W2 / Wo passes
  float* sR = sX + 16 * XSTR;             // dz2 (unmasked): residual into LayerNorm1's backward
  float* sC = sR + 16 * XSTR;             // staging / dgamma-dbeta reduction scratch (32 rows)
  float* sH = sC + 32 * XSTR;             // A0 tile first (dlogits or dqkv of the layer above), then dhid (16, F)
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, l16 = lane & 15, lg = lane >> 4;
  const bool cw = tid < 256;
  const int row0 = blockIdx.x * 16, d = a.d, F = a.F;
  const int grp = wave * 4 + lg, grow = row0 + grp;
  const bool live = grow < a.M;
  const size_t growc = live ? grow : 0;
  const float invN = 1.0f / (float)d;
  f32x4 acc[TN];
  float dg[CPL], db[CPL];
  // pass list: W0 (OutputLayer / in_proj of the layer above) | W2 in DPAD-column passes | W1 | Wo   (all read as [k][n])
  const int np1 = (F + NP - 1) / NP, npass = 3 + np1;
  auto desc_at = [&](int pi) -> PassDesc {
    if (pi == 0) return PassDesc{a.W0, d, a.K0, d, 0};
    if (pi <= np1) return PassDesc{a.W2, F, d, F, (pi - 1) * NP};
    if (pi == np1 + 1) return PassDesc{a.W1, d, F, d, 0};
    return PassDesc{a.Wo, d, d, d, 0};
  };
  const float* const zp = gt_zero_ptr();
  if (!cw) {
    // barriers the compute waves take after the last slab of a pass: accumulator staging (1) + 3 per dgamma/dbeta reduction
    const bool topl = a.gf != nullptr;
    auto nbar_at = [&](int pi) -> int { return pi == 0 ? (topl ? 7 : 4) : (pi == np1 + 1 ? 4 : 0); };
    chain_loader<TN, BK, true>(ring, npass, desc_at, nbar_at, tid - 256, zp);
    return;
  }
  int gslab = 0;
  for (int e = tid; e < 2 * 16 * XSTR + 32 * XSTR + 16 * HSTR; e += 256) work[e] = 0.f;
  GT_BARRIER();
  if (cw) chain_load_tile(sH, HSTR, a.A0, a.K0, row0, a.M, a.K0, tid, zp);

  // ---- stage 0: grad w.r.t. this layer's output, then (final norm's and) LayerNorm2's backward -> dz2, dz2m
  {
    float e_res[CPL], e_xh2[CPL], e_g2[CPL], e_xhf[CPL], e_gf[CPL], r2, rf;
    const bool top = a.gf != nullptr;
    r2 = 0.f; rf = 0.f;
    if (cw) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        const bool ok = live && c < d;
        chain_gload(e_res[i], a.res0, growc * d + c, ok && a.res0 != nullptr, zp);
        chain_gload(e_xh2[i], a.xhat2, growc * d + c, ok, zp); chain_gload(e_g2[i], a.g2, c, ok, zp);
        chain_gload(e_xhf[i], a.xhatf, growc * d + c, ok && top, zp); chain_gload(e_gf[i], a.gf, c, ok && top, zp);
      }
      chain_gload(r2, a.rstd2, growc, true, zp); chain_gload(rf, a.rstdf, growc, top, zp);
    }
    chain_pass<TN, BK, true>(acc, sH, HSTR, desc_at(0).K, ring, gslab, tid);
    if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    }
    GT_BARRIER();
    float dy[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) { dy[i] = 0.f; dg[i] = 0.f; db[i] = 0.f; }
    if (cw) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        dy[i] = (live && c < d) ? sC[grp * XSTR + c] + e_res[i] : 0.f;
      }
      if (top) chain_ln_bwd_row<CPL>(dy, e_xhf, e_gf, rf, live, d, l16, invN, dg, db);
    }
    if (top) chain_ln_partials<DPAD>(dg, db, sC, a.partf, d, tid, cw);
#pragma unroll
    for (int i = 0; i < CPL; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    if (cw) {
      chain_ln_bwd_row<CPL>(dy, e_xh2, e_g2, r2, live, d, l16, invN, dg, db);
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
    }
    chain_ln_partials<DPAD>(dg, db, sC, a.part2, d, tid, cw);
  }

  // ---- stage 1: dhid = (dz2m W2) * relu'/dropout mask (from the saved activation), kept in LDS and stored for wgrad
  {
    const int F16 = (F + 15) / 16 * 16;
    for (int p = 0; p < np1; ++p) {
      float hv[TN * 4];
#pragma unroll
      for (int t = 0; t < TN; ++t)
      if (cw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = p * NP + (wave * TN + t) * 16 + l16, row = row0 + 4 * lg + r;
          chain_gload(hv[t * 4 + r], a.hact, (size_t)row * F + col, col < F && row < a.M, zp);
        }
      }
      chain_pass<TN, BK, true>(acc, sX, XSTR, desc_at(1 + p).K, ring, gslab, tid);
      if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 4 * lg + r, col = p * NP + (wave * TN + t) * 16 + l16, row = row0 + rl;
          if (col < F) {
            const float v = (hv[t * 4 + r] != 0.f) ? acc[t][r] * a.mask_scale : 0.f;
            sH[rl * HSTR + col] = v;
            if (row < a.M) a.dhid_out[(size_t)row * F + col] = v;
          } else if (col < F16) {
            sH[rl * HSTR + col] = 0.f;
          }
        }
      }
    }
  }

  // ---- stage 2: dx1 = dhid W1 + dz2, LayerNorm1's backward -> dz1 (residual for the layer below), dz1m
  {
    float e_xh1[CPL], e_g1[CPL], r1 = 0.f;
    if (cw) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        const bool ok = live && c < d;
        chain_gload(e_xh1[i], a.xhat1, growc * d + c, ok, zp); chain_gload(e_g1[i], a.g1, c, ok, zp);
      }
      chain_gload(r1, a.rstd1, growc, true, zp);
    }
    chain_pass<TN, BK, true>(acc, sH, HSTR, desc_at(np1 + 1).K, ring, gslab, tid);
    if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(4 * lg + r) * XSTR + (wave * TN + t) * 16 + l16] = acc[t][r];
    }
    GT_BARRIER();
#pragma unroll
    for (int i = 0; i < CPL; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    if (cw) {
      float dy[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = l16 + 16 * i;
        dy[i] = (live && c < d) ? sC[grp * XSTR + c] + sR[grp * XSTR + c] : 0.f;
      }
      chain_ln_bwd_row<CPL>(dy, e_xh1, e_g1, r1, live, d, l16, invN, dg, db);
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
    }
    chain_ln_partials<DPAD>(dg, db, sC, a.part1, d, tid, cw);
  }

  // ---- stage 3: dctx = dz1m Wo (the attention backward kernel takes it from memory)
  {
    chain_pass<TN, BK, true>(acc, sX, XSTR, desc_at(np1 + 2).K, ring, gslab, tid);
    if (cw) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = (wave * TN + t) * 16 + l16, row = row0 + 4 * lg + r;
          if (col < d && row < a.M) a.dctx_out[(size_t)row * d + col] = acc[t][r];
        }
    }
  }
}
