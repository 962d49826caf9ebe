// Large-problem fp32 GEMM: 128x128 tiles on v_mfma_f32_32x32x2_f32 with a two-deep register prefetch ring and a compile-time
// interleave of staging instructions with the MFMAs.  Serves the interior-only, large-M forms of the Linear forward ("NT":
// B = weight (out,in), k contiguous) and dgrad ("NN": B = weight read as (k, n), n contiguous) with the plain / FFN1 / FFN2-dgrad
// epilogues; everything else (edges, small problems, K not a multiple of 64, the LayerNorm row tiles, weight gradients) stays
// on gt_gemm.h.  Results are bitwise those of any fp32 fmaf chain in k order 0,4,1,5,2,6,3,7 per 8-k group (the lane map below),
// i.e. the same numbers as gt_gemm.h's kernels up to the order of the fp32 additions.
//
// Why this shape (measured on the d_model 512 QKV projection, 16384 tokens; tools/ubench/gemm32_bench.hip has every step):
//   * 32x32x2 MFMA: lane (r = l & 31, h = l >> 5) reads ONE float4 (k = 4h..4h+3 of an 8-k group) per 32-row fragment, and with
//     that lane -> row map a row stride of BK + 4 floats is conflict-free for ds_read_b128 (the 16x16x4 map needs BK + 8), so a
//     double-buffered 128x128x32 workgroup takes 72 KB of LDS and two of them share a CU with room to spare.
//   * what the plain one-deep loop loses is neither LDS reads nor barriers (ablations: +-0) but the global -> LDS staging: with
//     loads issued at the top of a slab and consumed in its middle only ~32 KB per CU are in flight on average, and
//     37 GB/s per CU at 1-1.5 us loaded latency needs ~55 KB (Little).  The ring issues slab t+2 at the top of slab t and
//     writes slab t+1 to LDS in the middle of slab t: 75.7 % -> 82.4 % of the 157.3 TF peak.
//   * sched_group_barrier puts ONE staging instruction (global load / ds_write / ds_read) behind each MFMA instead of blocks
//     of 8 that hold the wave's issue port while the matrix pipe drains: 82.4 % -> 85.9 % (135 TF).
#pragma once
// (included at the end of gt_gemm.h: GemmArgs, gemm_label and the EPI_* constants come from there)

// n times: one MFMA, then one instruction of class `mask` (0x20 VMEM read, 0x100 DS read, 0x200 DS write)
#define GT_IL(mask, n) _Pragma("unroll") for (int z_ = 0; z_ < (n); ++z_) { GT_SGB(0x8, 1) GT_SGB(mask, 1) }

struct Gemm32Cfg {
  static constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
  // operand images keep the SOURCE orientation: k-contiguous source -> [row][BK + 4] (fragment = one ds_read_b128),
  // row-contiguous source (KM) -> [k][128 + 4] (fragment = four ds_read_b32, conflict-free: 32 consecutive floats per half wave)
  template <bool KM> static constexpr int str() { return KM ? 128 + 4 : BK + 4; }
  template <bool KM> static constexpr int sz() { return KM ? BK * (128 + 4) : 128 * (BK + 4); }
  template <bool AKM, bool BKM> static constexpr int smem() { return 2 * (sz<AKM>() + sz<BKM>()); }
};

// host side: can this problem go on the big-tile kernel?  (interior tiles only, even number of 32-wide slabs, 16-byte rows)
static inline bool gemm32_ok(const GemmArgs& g, int epi, bool bkm) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (g.M % 128 || g.N % 128 || g.K % 64 || g.K < 64) return false;
  if ((g.lda & 3) || (g.ldb & 3) || (g.ldc & 3) || !al16(g.A) || !al16(g.B) || !al16(g.C)) return false;
  if (epi == EPI_STORE || epi == EPI_RELU_DROP) { if (g.bias && !al16(g.bias)) return false; }
  if ((epi == EPI_MASK_NZ || epi == EPI_ADD_RELUMASK_DROP) && ((g.ldres & 3) || !al16(g.res) || (g.res16 && (reinterpret_cast<uintptr_t>(g.res16) & 7)))) return false;
  if (epi == EPI_ADD_RELUMASK_DROP && ((g.N & 3) || !al16(g.aux_in))) return false;
  (void)bkm;
  return true;
}

// The store epilogues of a tile of TA x TB 32x32 blocks per wave (128x128: 2 x 2; gt_gemm64.h: 1 x 1) (transposed product: lane (r32, h) holds ONE row of C per 32x32 tile and, in registers
// 4 q .. 4 q + 3, the four consecutive columns 8 q + 4 h + 0..3): two-phase -- every global input first, then compute + 16-byte stores.
// g.C16: a bf16 copy of the stored values as well (8-byte stores), for a consumer that takes this output as a GEMM operand.
// EPI_MASK_NZ with keep bits: the lane's words of its TA x TB blocks, requested BEFORE the main loop (2 registers on the big tile) -- the
// epilogue then starts on values that arrived long ago instead of on a round trip (72.6 -> 7x us per FFN2 dgrad at d_model 512 / 16384 tokens)
template <int EPI, int TA, int TB>
__device__ __forceinline__ void gemm32_kbits_pre(const GemmArgs& g, const int m0, const int n0, const int wm, const int wn, const int r32, const int h,
                                                 uint32_t (&kbpre)[TA * TB]) {
#pragma unroll
  for (int i = 0; i < TA * TB; ++i) kbpre[i] = 0u;
  if constexpr (EPI == EPI_MASK_NZ) {
    if (g.kbits != nullptr) {
#pragma unroll
      for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
          const int bi = (m0 + wm * (32 * TA) + ta * 32) >> 5, bj = (n0 + wn * (32 * TB) + tb * 32) >> 5;
          kbpre[ta * TB + tb] = g.kbits[((size_t)bi * (g.N >> 5) + bj) * 64 + 32 * h + r32];
        }
    }
  }
}
template <int EPI, int TA = 2, int TB = 2>
__device__ __forceinline__ void gemm32_store_epilogue(const GemmArgs& g, const f32x16 (&acc)[TA][TB], const int m0, const int n0, const int wm, const int wn,
                                                      const int r32, const int h, const uint32_t* kbpre = nullptr) {
  const uint32_t dkey = gt_drop_key(g.drop);
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    f32x4 bia[4], rin[TA][4], rin2[TA][4];
    // keep bits of the FFN activation (g.kbits: EPI_RELU_DROP writes them, EPI_MASK_NZ reads them in place of the activation itself -- 1 bit
    // for 32 or 16): one 16-bit word per (32x32 block of the [M][N] output, lane of this layout) -- block (row / 32, col / 32), word
    // 32 h + r32 of the block's 64, bit 4 q4 + r <-> column 8 q4 + 4 h + r of the block: a function of (row, col) alone, so the 128x128
    // and the 64x64 tile, fp32 and bf16 sources agree on it
    uint32_t kb[TA];
    uint16_t* kbp[TA];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta) {
      kb[ta] = 0u; kbp[ta] = nullptr;
      if ((EPI == EPI_MASK_NZ || EPI == EPI_RELU_DROP) && g.kbits != nullptr) {
        const int bi = (m0 + wm * (32 * TA) + ta * 32) >> 5, bj = (n0 + wn * (32 * TB) + tb * 32) >> 5;
        kbp[ta] = g.kbits + ((size_t)bi * (g.N >> 5) + bj) * 64 + 32 * h + r32;
        if (EPI == EPI_MASK_NZ) kb[ta] = kbpre != nullptr ? kbpre[ta * TB + tb] : (uint32_t)*kbp[ta];
      }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int col = n0 + wn * (32 * TB) + tb * 32 + 8 * q4 + 4 * h;
      bia[q4] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((EPI == EPI_STORE || EPI == EPI_RELU_DROP) && g.bias != nullptr) bia[q4] = *reinterpret_cast<const f32x4*>(g.bias + col);
#pragma unroll
      for (int ta = 0; ta < TA; ++ta) {
        const int row = m0 + wm * (32 * TA) + ta * 32 + r32;
        rin[ta][q4] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_STORE && g.accumulate) rin[ta][q4] = *reinterpret_cast<const f32x4*>(g.C + (size_t)row * g.ldc + col);
#if defined(GT_ACCT_NOMASK)
        if (EPI == EPI_MASK_NZ) { rin[ta][q4] = f32x4{1.f, 1.f, 1.f, 1.f}; continue; }    // (measurement build, WRONG RESULTS: the FFN2 dgrad without its mask read)
#endif
        if (EPI == EPI_MASK_NZ && g.kbits != nullptr) continue;       // (the mask comes as bits: kb below)
        if (EPI == EPI_MASK_NZ && g.res16 != nullptr) {      // the mask source (hact) lives in bf16 only: zero / non-zero is all that is asked
          const uint2 hb = *reinterpret_cast<const uint2*>(g.res16 + (size_t)row * g.ldres + col);
          rin[ta][q4] = f32x4{gt_u2f(hb.x << 16), gt_u2f(hb.x & 0xFFFF0000u), gt_u2f(hb.y << 16), gt_u2f(hb.y & 0xFFFF0000u)};
        } else
        if (EPI == EPI_MASK_NZ || EPI == EPI_ADD_RELUMASK_DROP) rin[ta][q4] = *reinterpret_cast<const f32x4*>(g.res + (size_t)row * g.ldres + col);
        if (EPI == EPI_ADD_RELUMASK_DROP) rin2[ta][q4] = *reinterpret_cast<const f32x4*>(g.aux_in + (size_t)row * g.N + col);
      }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int col = n0 + wn * (32 * TB) + tb * 32 + 8 * q4 + 4 * h;
#pragma unroll
      for (int ta = 0; ta < TA; ++ta) {
        const int row = m0 + wm * (32 * TA) + ta * 32 + r32;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[ta][tb][4 * q4 + r];
          if (EPI == EPI_STORE) v = v + bia[q4][r] + rin[ta][q4][r];
          else if (EPI == EPI_RELU_DROP) v = fmaxf(v + bia[q4][r], 0.f) * gt_drop_mul(g.drop, dkey, (uint32_t)(row * g.N + col + r));
          else if (EPI == EPI_MASK_NZ) v = ((g.kbits != nullptr ? ((kb[ta] >> (4 * q4 + r)) & 1u) != 0u : rin[ta][q4][r] != 0.f)) ? v * g.mask_scale : 0.f;
          else if (EPI == EPI_ADD_RELUMASK_DROP) {
            v = (v + rin[ta][q4][r]) * gt_drop_mul(g.drop, dkey, (uint32_t)(row * g.N + col + r));
            v = (rin2[ta][q4][r] > 0.f) ? v : 0.f;
          }
          o[r] = v;
          // (the bit a consumer would derive from the stored value: of the bf16 copy when that is the only one stored)
          if (EPI == EPI_RELU_DROP && g.kbits != nullptr) kb[ta] |= (uint32_t)(g.C != nullptr ? v != 0.f : (gt_f2bf(v) & 0x7FFFu) != 0u) << (4 * q4 + r);
        }
#if defined(GT_EPI_NOSTORE)
        asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));            // (measurement build: the main loop and the epilogue's arithmetic without its stores)
        continue;
#endif
        if (g.C != nullptr) *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = o;      // (nullptr: the output lives in bf16 only)
        if (g.C16 != nullptr) {
          uint2 pk;
          pk.x = (uint32_t)gt_f2bf(o[0]) | ((uint32_t)gt_f2bf(o[1]) << 16); pk.y = (uint32_t)gt_f2bf(o[2]) | ((uint32_t)gt_f2bf(o[3]) << 16);
          *reinterpret_cast<uint2*>(g.C16 + (size_t)row * g.ldc16 + col) = pk;
        }
      }
    }
    if (EPI == EPI_RELU_DROP && g.kbits != nullptr) {
#pragma unroll
      for (int ta = 0; ta < TA; ++ta) *kbp[ta] = (uint16_t)kb[ta];
    }
  }
}

// LayerNorm epilogues on this tile through the in-launch row exchange (round 6; defined in gt_gemm64.h beside the 64x64 form)
template <int EPI, int NPH>
__device__ __forceinline__ void gemm32_ln_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], const int m0, const int n0, const int wm, const int wn,
                                                   const int r32, const int h, const uint32_t (&seq)[2], float* smem);
__device__ __forceinline__ void g128_seq(const GemmArgs& g, const int m0, const int n0, const int wm, const int wn, const int r32, const int h, uint32_t (&seq)[2]);

// One 128x128 output tile: rows m0.., columns n0.., contraction range [kbeg, kbeg + nk * 32) (nk even, >= 2).
// AKM / BKM as in gt_gemm.h: operand stored with the contraction index as its ROW index (weight gradients: both; dgrad: B).
// PREC = 1 (gt_config.precision = 1): the same staging, fp32 LDS images in source orientation; the operands are rounded to bf16
// when a lane assembles its fragment (8 consecutive k per lane half, two ds_read_b128 or eight ds_read_b32) and go through
// v_mfma_f32_32x32x16_bf16.  16x fewer matrix cycles: the loop is then bound by the global -> LDS staging alone, so it runs a
// plain one-barrier-per-slab schedule on the same two-deep ring.
// SA16 / SB16 (weight gradients at PREC = 1 only: both operands token-major): that operand is staged from its bf16 SHADOW (g.A16 / g.B16,
// row stride lda16 / ldb16 elements) -- 16-byte loads of 8 elements, widened to fp32 on their way into the SAME fp32 LDS image (a bf16
// value is its fp32 neighbour's upper half), so everything behind the staging, and every result, is unchanged: half the bytes fetched.
typedef f32x4 __attribute__((may_alias)) f32x4_raw;          // 16 raw bytes (8 bf16) held in a float4 register set
template <bool AKM, bool BKM, int EPI, int PREC = 0, bool SA16 = false, bool SB16 = false>
__device__ __forceinline__ void gemm32_body(const GemmArgs& g, const int m0, const int n0, const int kbeg, const int nk, const bool want_dbias,
                                            float* smem) {
  static_assert((!SA16 && !SB16) || (AKM && BKM && PREC == 1), "bf16 sources: the weight-gradient form at precision 1");
  typedef Gemm32Cfg Cfg;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
  constexpr int SA_STR = Cfg::str<AKM>(), SA_SZ = Cfg::sz<AKM>(), SB_STR = Cfg::str<BKM>(), SB_SZ = Cfg::sz<BKM>();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  uint32_t xtag[2] = {0u, 0u};   // LayerNorm epilogues: this launch's sequence numbers = the lane's OWN granules' + 1 (read before anything is published)
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) g128_seq(g, m0, n0, wm, wn, r32, h, xtag);
  uint32_t kbpre[4];
  gemm32_kbits_pre<EPI, 2, 2>(g, m0, n0, wm, wn, r32, h, kbpre);

  // staging: 4 float4 per thread and operand per slab.  A: chunk (row, 4 k);  B: chunk (n, 4 k) or, BKM, (k, 4 n)
  constexpr int PER = 4;
  f32x4 va[PER], vb[PER], wa[PER], wb[PER];
  const char* pa[PER];
  const char* pb[PER];
  int sa_off[PER], sb_off[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256;
    const int r = ch >> 3, c = (ch & 7) * 4;                   // k-contiguous source: (row, 4 k)
    const int kr = ch >> 5, cn = (ch & 31) * 4;                // row-contiguous source: (k, 4 rows)
    if (AKM) {
      pa[i] = reinterpret_cast<const char*>(g.A + (size_t)(kbeg + kr) * g.lda + m0 + cn);
      sa_off[i] = kr * SA_STR + cn;
    } else {
      pa[i] = reinterpret_cast<const char*>(g.A + (size_t)(m0 + r) * g.lda + kbeg + c);
      sa_off[i] = r * SA_STR + c;
    }
    if (BKM) {
      pb[i] = reinterpret_cast<const char*>(g.B + (size_t)(kbeg + kr) * g.ldb + n0 + cn);
      sb_off[i] = kr * SB_STR + cn;
    } else {
      pb[i] = reinterpret_cast<const char*>(g.B + (size_t)(n0 + r) * g.ldb + kbeg + c);
      sb_off[i] = r * SB_STR + c;
    }
    // bf16 sources (row-contiguous only): chunk = (k, 8 rows) = 16 bytes; PER / 2 chunks per thread and slab, sets [0 .. PER / 2) in use
    if (i < PER / 2) {
      const int kr16 = ch >> 4, cn16 = (ch & 15) * 8;
      if (SA16) { pa[i] = reinterpret_cast<const char*>(g.A16 + (size_t)(kbeg + kr16) * g.lda16 + m0 + cn16); sa_off[i] = kr16 * SA_STR + cn16; }
      if (SB16) { pb[i] = reinterpret_cast<const char*>(g.B16 + (size_t)(kbeg + kr16) * g.ldb16 + n0 + cn16); sb_off[i] = kr16 * SB_STR + cn16; }
    }
  }
  const size_t astep = SA16 ? (size_t)g.lda16 * 2 : AKM ? (size_t)g.lda * 4 : 4, bstep = SB16 ? (size_t)g.ldb16 * 2 : BKM ? (size_t)g.ldb * 4 : 4;    // bytes per k
  // 8 bf16 in a register set -> two float4: element 2 w is the low half of word w, 2 w + 1 its high half
  auto widen_lo = [](const f32x4& v) { return f32x4{gt_u2f(gt_f2u(v[0]) << 16), gt_u2f(gt_f2u(v[0]) & 0xFFFF0000u), gt_u2f(gt_f2u(v[1]) << 16), gt_u2f(gt_f2u(v[1]) & 0xFFFF0000u)}; };
  auto widen_hi = [](const f32x4& v) { return f32x4{gt_u2f(gt_f2u(v[2]) << 16), gt_u2f(gt_f2u(v[2]) & 0xFFFF0000u), gt_u2f(gt_f2u(v[3]) << 16), gt_u2f(gt_f2u(v[3]) & 0xFFFF0000u)}; };
#define G32_LD(XA, XB, k0)                                                                     \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    if (!SA16 || i < PER / 2) XA[i] = *reinterpret_cast<const f32x4_raw*>(pa[i] + (size_t)(k0) * astep); \
    if (!SB16 || i < PER / 2) XB[i] = *reinterpret_cast<const f32x4_raw*>(pb[i] + (size_t)(k0) * bstep); \
  }
#define G32_ST(XA, XB, buf)                                                                    \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    if (!SA16) *reinterpret_cast<f32x4*>(&smem[(buf) * SA_SZ + sa_off[i]]) = XA[i];            \
    else if (i < PER / 2) {                                                                    \
      *reinterpret_cast<f32x4*>(&smem[(buf) * SA_SZ + sa_off[i]]) = widen_lo(XA[i]);           \
      *reinterpret_cast<f32x4*>(&smem[(buf) * SA_SZ + sa_off[i] + 4]) = widen_hi(XA[i]);       \
    }                                                                                          \
    if (!SB16) *reinterpret_cast<f32x4*>(&smem[2 * SA_SZ + (buf) * SB_SZ + sb_off[i]]) = XB[i]; \
    else if (i < PER / 2) {                                                                    \
      *reinterpret_cast<f32x4*>(&smem[2 * SA_SZ + (buf) * SB_SZ + sb_off[i]]) = widen_lo(XB[i]); \
      *reinterpret_cast<f32x4*>(&smem[2 * SA_SZ + (buf) * SB_SZ + sb_off[i] + 4]) = widen_hi(XB[i]); \
    }                                                                                          \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // fragments of one 8-k group: element j of lane half h is k = 4h + j of the group, for A and B alike
  f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
  const int offa = AKM ? (4 * h) * SA_STR + wm * 64 + r32 : (wm * 64 + r32) * SA_STR + 4 * h;
  const int offb = 2 * SA_SZ + (BKM ? (4 * h) * SB_STR + wn * 64 + r32 : (wn * 64 + r32) * SB_STR + 4 * h);
#define G32_RD(FA, FB, buf, kk)                                                                \
  if (AKM) {                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
      FA[0][j] = smem[(buf) * SA_SZ + offa + ((kk) * 8 + j) * SA_STR];                         \
      FA[1][j] = smem[(buf) * SA_SZ + offa + ((kk) * 8 + j) * SA_STR + 32];                    \
    }                                                                                          \
  } else {                                                                                     \
    FA[0] = *reinterpret_cast<const f32x4*>(smem + (buf) * SA_SZ + offa + (kk) * 8);           \
    FA[1] = *reinterpret_cast<const f32x4*>(smem + (buf) * SA_SZ + offa + 32 * SA_STR + (kk) * 8); \
  }                                                                                            \
  if (BKM) {                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
      FB[0][j] = smem[(buf) * SB_SZ + offb + ((kk) * 8 + j) * SB_STR];                         \
      FB[1][j] = smem[(buf) * SB_SZ + offb + ((kk) * 8 + j) * SB_STR + 32];                    \
    }                                                                                          \
  } else {                                                                                     \
    FB[0] = *reinterpret_cast<const f32x4*>(smem + (buf) * SB_SZ + offb + (kk) * 8);           \
    FB[1] = *reinterpret_cast<const f32x4*>(smem + (buf) * SB_SZ + offb + 32 * SB_STR + (kk) * 8); \
  }
  // Store epilogues take the transposed product (first operand = B fragment): lane (r32, h) ends up with ONE row of C per
  // 32x32 tile and, in registers 4g..4g+3, the four consecutive columns 8g + 4h + 0..3 -> 16-byte accesses.  The atomic
  // epilogue keeps the plain product: register e of a tile is row (e&3) + 8(e>>2) + 4h, column r32 -- one atomic instruction
  // covers two 128-byte row segments, the full-rate shape (MI355X_MICROARCH.md, "Global float atomics").
#define G32_MM(FA, FB)                                                                         \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                \
  _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                             \
  _Pragma("unroll") for (int tb = 0; tb < 2; ++tb)                                             \
    acc[ta][tb] = (EPI != EPI_ATOMIC) ? GT_MFMA32(FB[tb][j], FA[ta][j], acc[ta][tb]) : GT_MFMA32(FA[ta][j], FB[tb][j], acc[ta][tb]);
  constexpr int NRD = (AKM ? 8 : 2) + (BKM ? 8 : 2), NRD8 = NRD > 8 ? 8 : NRD, NRDF = NRD > 16 ? 16 : NRD;  // LDS reads per fragment group
  float bsum = 0.f;                                            // EPI_ATOMIC: bias-gradient partial (column sum of the A slabs)
  G32_LD(va, vb, 0)
  G32_LD(wa, wb, BK)
  G32_ST(va, vb, 0)
  __syncthreads();
  if constexpr (PREC == 1) {
    // fragment of k-step s_ (16 k) of a 32-row tile: element j of lane half h is k = 16 s_ + 8 h + j
    auto frag = [&](const int base, const int str, const bool km, const int tile, const int s_) -> bf16x8 {
      bf16x8 r;
      if (km) {
#pragma unroll
        for (int j = 0; j < 8; ++j) GT_BF16X8_SET(r, j, smem[base + (16 * s_ + 8 * h + j) * str + tile * 32 + r32]);
      } else {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + base + (tile * 32 + r32) * str + 16 * s_ + 8 * h);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + base + (tile * 32 + r32) * str + 16 * s_ + 8 * h + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { GT_BF16X8_SET(r, j, lo[j]); GT_BF16X8_SET(r, 4 + j, hi[j]); }
      }
      return r;
    };
    const int abase = AKM ? wm * 64 : (wm * 64) * SA_STR, bbase = 2 * SA_SZ + (BKM ? wn * 64 : (wn * 64) * SB_STR);
#define G32_SLAB16(CUR, NA, NB, FA_, FB_, t)                                                   \
    { const int k2_ = ((t) + 2 < nk ? (t) + 2 : nk - 1) * BK;                                  \
      G32_LD(FA_, FB_, k2_) }                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                         \
      bf16x8 a16[2], b16[2];                                                                   \
      _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) {                                       \
        a16[ti] = frag((CUR) * SA_SZ + abase, SA_STR, AKM, ti, s_);                            \
        b16[ti] = frag((CUR) * SB_SZ + bbase, SB_STR, BKM, ti, s_);                            \
      }                                                                                        \
      _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                         \
      _Pragma("unroll") for (int tb = 0; tb < 2; ++tb)                                         \
        acc[ta][tb] = (EPI != EPI_ATOMIC) ? GT_MFMA32_BF16(b16[tb], a16[ta], acc[ta][tb]) : GT_MFMA32_BF16(a16[ta], b16[tb], acc[ta][tb]); \
    }                                                                                          \
    if (EPI == EPI_ATOMIC && AKM && want_dbias && tid < BM) {  /* column sums of the bf16-rounded dY slab */ \
      _Pragma("unroll 8") for (int kk_ = 0; kk_ < BK; ++kk_) bsum += gt_bf2f(gt_f2bf(smem[(CUR) * SA_SZ + kk_ * SA_STR + tid])); \
    }                                                                                          \
    G32_ST(NA, NB, (CUR) ^ 1)                                                                  \
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      G32_SLAB16(0, wa, wb, va, vb, kt)
      G32_SLAB16(1, va, vb, wa, wb, kt + 1)
    }
#undef G32_SLAB16
  } else {
  G32_RD(fa0, fb0, 0, 0)
  // one slab.  CUR: LDS buffer holding slab t; (NA, NB): registers holding slab t+1; (FA_, FB_): the set slab t came from,
  // free again -> receives slab t+2 (the last two slabs re-load the final slab: branch-free, never used)
#define G32_SLAB(CUR, NA, NB, FA_, FB_, t)                                                     \
  { const int k2_ = ((t) + 2 < nk ? (t) + 2 : nk - 1) * BK;                                    \
    G32_LD(FA_, FB_, k2_) }                                                                    \
  G32_RD(fa1, fb1, CUR, 1) G32_MM(fa0, fb0)                                                    \
  GT_IL(0x20, 8) GT_IL(0x100, NRD8) GT_SGB(0x8, 16 - 8 - NRD8)                                 \
  GT_SCHED_FENCE()                                                                             \
  G32_RD(fa0, fb0, CUR, 2) G32_MM(fa1, fb1)                                                    \
  GT_IL(0x100, NRDF) GT_SGB(0x8, 16 - NRDF)                                                    \
  GT_SCHED_FENCE()                                                                             \
  G32_ST(NA, NB, (CUR) ^ 1)                                                                    \
  G32_RD(fa1, fb1, CUR, 3) G32_MM(fa0, fb0)                                                    \
  GT_IL(0x200, 8) GT_IL(0x100, NRD8) GT_SGB(0x8, 16 - 8 - NRD8)                                \
  GT_SCHED_FENCE()                                                                             \
  /* bias gradient: every read of buffer CUR must precede this slab's barrier (the next slab overwrites it after it) */ \
  if (EPI == EPI_ATOMIC && AKM && want_dbias && tid < BM) {                                    \
    _Pragma("unroll 8") for (int kk_ = 0; kk_ < BK; ++kk_) bsum += smem[(CUR) * SA_SZ + kk_ * SA_STR + tid]; \
  }                                                                                            \
  __syncthreads();                                                                             \
  G32_RD(fa0, fb0, (CUR) ^ 1, 0)                                                               \
  G32_MM(fa1, fb1)                                                                             \
  GT_IL(0x100, NRDF) GT_SGB(0x8, 16 - NRDF)                                                    \
  GT_SCHED_FENCE()
  for (int kt = 0; kt < nk; kt += 2) {
    G32_SLAB(0, wa, wb, va, vb, kt)
    G32_SLAB(1, va, vb, wa, wb, kt + 1)
  }
  }   // PREC

  if (EPI == EPI_ATOMIC) {
    if (AKM && want_dbias && tid < BM) atomicAdd(&g.dbias[m0 + tid], bsum);
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 64 + ta * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn * 64 + tb * 32 + r32;
          atomicAdd(&g.C[(size_t)row * g.ldc + col], acc[ta][tb][e]);
        }
    return;
  }
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) {
    if (g.N == 512) gemm32_ln_epilogue<EPI, 4>(g, acc, m0, n0, wm, wn, r32, h, xtag, smem);
    else            gemm32_ln_epilogue<EPI, 2>(g, acc, m0, n0, wm, wn, r32, h, xtag, smem);
  } else {
    gemm32_store_epilogue<EPI>(g, acc, m0, n0, wm, wn, r32, h, kbpre);
  }
#undef G32_LD
#undef G32_ST
#undef G32_RD
#undef G32_MM
#undef G32_SLAB
}

// ================================================================================================================ bf16 SOURCES
// gt_config.precision = 1 at sizes where the loop above is bound by the global -> LDS staging (fp32 sources: 32 KB per 128x128x32
// slab): both operands come as bf16 SHADOWS (g.A16 [M][K], g.B16 [N][K], k contiguous: activations written by their producers next to
// the fp32 tensors, weights by weight_shadow_kernel -- the transposed copy makes a dgrad this same NT form).  Same bytes per slab, twice
// the k: 128x128x64 slabs, bf16 LDS images [128][64 + 8] (144-byte rows: a fragment = one ds_read_b128, conflict-free), a three-deep
// register ring, v_mfma_f32_32x32x16_bf16 with the k -> (lane half, element) map of the PREC = 1 body above: the sums are the same
// numbers.  Measured in isolation (round 3, tools/rejected/gemm32h_bf16_shadows.md): 1.28-1.34x the fp32-source kernel at K = 512.
struct Gemm32hCfg {
  static constexpr int BK = 64, STR = BK + 8, SZ = 128 * STR;           // bf16 elements
};
#ifdef GT_EMU
struct __attribute__((may_alias, aligned(16))) G32hRegs { uint32_t v[4]; };       // 16 bytes = 8 bf16
#else
typedef uint32_t G32hRegs __attribute__((ext_vector_type(4)));                    // (one global_load_dwordx4 / ds_write_b128 each)
#endif
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm32h_kernel(GemmArgs g) {
  typedef Gemm32hCfg Cfg;
  constexpr int BK = Cfg::BK, STR = Cfg::STR, SZ = Cfg::SZ, PER = 4;
  __shared__ __attribute__((aligned(16))) uint16_t sm[4 * SZ];           // [buffer][A | B]
  const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const int xcd = lin & 7, q = nb >> 3, rr = nb & 7;
  const int bid = xcd * q + (xcd < rr ? xcd : rr) + (lin >> 3);
  const int m0 = (bid / gx) * 128, n0 = (bid % gx) * 128, nk = g.K / BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  uint32_t xtag[2] = {0u, 0u};
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) g128_seq(g, m0, n0, wm, wn, r32, h, xtag);
  uint32_t kbpre[4];
  gemm32_kbits_pre<EPI, 2, 2>(g, m0, n0, wm, wn, r32, h, kbpre);
  G32hRegs a0[PER], b0[PER], a1[PER], b1[PER], a2[PER], b2[PER];
  const uint16_t* pa[PER];
  const uint16_t* pb[PER];
  int so[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch >> 3, c = (ch & 7) * 8;         // (row, 8 k) = 16 bytes
    pa[i] = g.A16 + (size_t)(m0 + r) * g.lda16 + c;
    pb[i] = g.B16 + (size_t)(n0 + r) * g.ldb16 + c;
    so[i] = r * STR + c;
  }
#define G32H_LD(XA, XB, k0)                                                                    \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    XA[i] = *reinterpret_cast<const G32hRegs*>(pa[i] + (k0));                                  \
    XB[i] = *reinterpret_cast<const G32hRegs*>(pb[i] + (k0));                                  \
  }
#define G32H_ST(XA, XB, buf)                                                                   \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    *reinterpret_cast<G32hRegs*>(&sm[(buf) * 2 * SZ + so[i]]) = XA[i];                         \
    *reinterpret_cast<G32hRegs*>(&sm[(buf) * 2 * SZ + SZ + so[i]]) = XB[i];                    \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int fa = (wm * 64 + r32) * STR + 8 * h, fb = SZ + (wn * 64 + r32) * STR + 8 * h;
  // a three-deep register ring: the A panel comes from HBM once per 128 rows and every slab waits for its slowest load
  auto kof = [&](const int t) { return (t < nk ? t : nk - 1) * BK; };
  G32H_LD(a0, b0, 0)
  G32H_LD(a1, b1, kof(1))
  G32H_LD(a2, b2, kof(2))
  G32H_ST(a0, b0, 0)
  G32H_LD(a0, b0, kof(3))
  __syncthreads();
  // slab t from LDS buffer CUR; (NA, NB) hold slab t + 1: written to the other buffer, then reloaded with slab t + 4
#define G32H_SLAB(CUR, NA, NB, t)                                                              \
  if ((t) < nk) {                                                                              \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                           \
    bf16x8 a16[2], b16[2];                                                                     \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) {                                         \
      a16[ti] = *reinterpret_cast<const bf16x8*>(&sm[(CUR) * 2 * SZ + fa + ti * 32 * STR + 16 * s_]); \
      b16[ti] = *reinterpret_cast<const bf16x8*>(&sm[(CUR) * 2 * SZ + fb + ti * 32 * STR + 16 * s_]); \
    }                                                                                          \
    _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                           \
    _Pragma("unroll") for (int tb = 0; tb < 2; ++tb)                                           \
      acc[ta][tb] = GT_MFMA32_BF16(b16[tb], a16[ta], acc[ta][tb]);                             \
  }                                                                                            \
  G32H_ST(NA, NB, (CUR) ^ 1)                                                                   \
  G32H_LD(NA, NB, kof((t) + 4))                                                                \
  __syncthreads();                                                                             \
  }
  for (int kt = 0; kt < nk; kt += 6) {
    G32H_SLAB(0, a1, b1, kt) G32H_SLAB(1, a2, b2, kt + 1) G32H_SLAB(0, a0, b0, kt + 2)
    G32H_SLAB(1, a1, b1, kt + 3) G32H_SLAB(0, a2, b2, kt + 4) G32H_SLAB(1, a0, b0, kt + 5)
  }
#undef G32H_SLAB
#undef G32H_LD
#undef G32H_ST
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) {
    float* const fsm = reinterpret_cast<float*>(sm);             // (4 x 128 x 72 bf16 = 73728 bytes: room for the partials' 34 KB)
    if (g.N == 512) gemm32_ln_epilogue<EPI, 4>(g, acc, m0, n0, wm, wn, r32, h, xtag, fsm);
    else            gemm32_ln_epilogue<EPI, 2>(g, acc, m0, n0, wm, wn, r32, h, xtag, fsm);
  } else {
    gemm32_store_epilogue<EPI>(g, acc, m0, n0, wm, wn, r32, h, kbpre);
  }
}
// host side: shadows present, interior tiles, 16-byte rows
static inline bool gemm32h_ok(const GemmArgs& g, int epi) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!g.A16 || !g.B16 || (g.lda16 & 7) || (g.ldb16 & 7) || !al16(g.A16) || !al16(g.B16)) return false;
  if (g.C16 && ((g.ldc16 & 3) || (reinterpret_cast<uintptr_t>(g.C16) & 7))) return false;
  if (g.accumulate) return false;
  GemmArgs t = g; t.A = reinterpret_cast<const float*>(g.A16); t.B = reinterpret_cast<const float*>(g.B16); t.lda = t.ldb = 4;
  return gemm32_ok(t, epi, false);
}
template <bool BKM, int EPI>
static inline void gemm32h_launch(const GemmArgs& g, hipStream_t s) {
  gt_prof_tag(gemm_label<BKM, EPI>(), 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + 4.0 * (double)g.M * g.N);
  gt_launch(gemm32h_kernel<EPI>, dim3(g.N / 128, g.M / 128), dim3(256), s, g);
}

// ================================================================================================================ row-owning tiles
// Round 3: the LayerNorm-fused Linears (EPI_RES_LN / EPI_RES_LNBWD of gt_gemm.h: the workgroup owns WHOLE rows) on the ring body.
// The one-deep 16x16x4 row tiles ran at 35 % (d_model 256) / 52-65 % (512) of the MFMA peak: 256 VGPR + 130 AGPR at one wave per
// SIMD, no prefetch distance.  Here: v_mfma_f32_32x32x2_f32, 8 waves (2 per SIMD), BMW x (8 / BMW) waves of 32 x (32 NB) columns,
// BK = 16 slabs double-buffered in LDS with the next-but-one slab in registers (the two-deep ring of gemm32_body), then the
// accumulators go through LDS ([BM][BN + 4], over the operand buffers) into gemm_row_epilogue -- the same arithmetic, in the same
// order per row, as the tiles they replace.  BN = d_model (256 or 512), interior only: M % BM == 0, K % 32 == 0, 16-byte rows.
template <int BN, int BMW> struct Gemm32RowCfg {
  static constexpr int BM = 32 * BMW, BK = 16, NT = 512, WN = 8 / BMW, NB = BN / (32 * WN);      // NB 32-column fragments per wave
  template <bool KM> static constexpr int bstr() { return KM ? BN + 4 : BK + 4; }
  template <bool KM> static constexpr int bsz() { return KM ? BK * (BN + 4) : BN * (BK + 4); }
  static constexpr int ASTR = BK + 4, ASZ = BM * ASTR;
  template <bool KM> static constexpr int main_sz() { return 2 * (ASZ + bsz<KM>()); }
  // epilogue: the [BM][BN + 4] accumulator image; the LayerNorm backward re-uses it for 2 x (NT / 16) rows of dgamma / dbeta partials
  template <int EPI> static constexpr int epi_sz() { return ((EPI == EPI_RES_LNBWD && 2 * (NT / 16) > BM) ? 2 * (NT / 16) : BM) * (BN + 4); }
  template <bool KM, int EPI> static constexpr int smem() { return main_sz<KM>() > epi_sz<EPI>() ? main_sz<KM>() : epi_sz<EPI>(); }
};
template <int BN, int BMW, bool BKM, int EPI>
__global__ __launch_bounds__(512, 1) void gemm32row_kernel(GemmArgs g) {
  typedef Gemm32RowCfg<BN, BMW> Cfg;
  constexpr int BM = Cfg::BM, BK = Cfg::BK, NT = Cfg::NT, WN = Cfg::WN, NB = Cfg::NB;
  constexpr int ASTR = Cfg::ASTR, ASZ = Cfg::ASZ, BSTR = Cfg::template bstr<BKM>(), BSZ = Cfg::template bsz<BKM>();
  __shared__ __attribute__((aligned(16))) float smem[Cfg::template smem<BKM, EPI>()];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int r32 = lane & 31, h = lane >> 5;
  const int by = blockIdx.x, m0 = by * BM;
  // staging: A = BM x 16 floats per slab (BM * 4 float4: the first BM * 4 threads take one each); B = BN x 16 floats = BN * 4 float4
  constexpr int PB = BN * 4 / NT;                              // float4 of B per thread and slab (2 at 256, 4 at 512)
  const bool has_a = tid < BM * 4;
  const char* pa = reinterpret_cast<const char*>(g.A + (size_t)(m0 + (has_a ? tid >> 2 : 0)) * g.lda + (tid & 3) * 4);
  const int sa_off = (tid >> 2) * ASTR + (tid & 3) * 4;
  const char* pb[PB];
  int sb_off[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int ch = tid + i * NT;
    if (BKM) { const int kr = ch / (BN / 4), cn = (ch % (BN / 4)) * 4; pb[i] = reinterpret_cast<const char*>(g.B + (size_t)kr * g.ldb + cn); sb_off[i] = kr * BSTR + cn; }
    else { const int r = ch >> 2, c = (ch & 3) * 4; pb[i] = reinterpret_cast<const char*>(g.B + (size_t)r * g.ldb + c); sb_off[i] = r * BSTR + c; }
  }
  const size_t bstep = BKM ? (size_t)g.ldb * 4 : 4;            // bytes per k
  f32x4 va, wa, vb[PB], wb[PB];
#define G32R_LD(XA, XB, k0)                                                                    \
  XA = has_a ? *reinterpret_cast<const f32x4*>(pa + (size_t)(k0) * 4) : f32x4{0.f, 0.f, 0.f, 0.f}; \
  _Pragma("unroll") for (int i = 0; i < PB; ++i) XB[i] = *reinterpret_cast<const f32x4*>(pb[i] + (size_t)(k0) * bstep);
#define G32R_ST(XA, XB, buf)                                                                   \
  if (has_a) *reinterpret_cast<f32x4*>(&smem[(buf) * ASZ + sa_off]) = XA;                      \
  _Pragma("unroll") for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(&smem[2 * ASZ + (buf) * BSZ + sb_off[i]]) = XB[i];
  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int offa = (wm * 32 + r32) * ASTR + 4 * h;
  const int offb = 2 * ASZ + (BKM ? (4 * h) * BSTR + wn * (32 * NB) + r32 : (wn * (32 * NB) + r32) * BSTR + 4 * h);
  // one slab = two 8-k groups; fragment element j of lane half h is k = 4 h + j of the group, for A and B alike (gemm32_body)
#define G32R_SLAB(CUR)                                                                         \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                           \
    const f32x4 fa = *reinterpret_cast<const f32x4*>(smem + (CUR) * ASZ + offa + kk * 8);      \
    f32x4 fb[NB];                                                                              \
    _Pragma("unroll") for (int t = 0; t < NB; ++t) {                                           \
      if (BKM) { _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[t][j] = smem[(CUR) * BSZ + offb + (kk * 8 + j) * BSTR + 32 * t]; } \
      else fb[t] = *reinterpret_cast<const f32x4*>(smem + (CUR) * BSZ + offb + (32 * t) * BSTR + kk * 8); \
    }                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                              \
    _Pragma("unroll") for (int t = 0; t < NB; ++t) acc[t] = GT_MFMA32(fb[t][j], fa[j], acc[t]); \
  }
  const int nk = g.K / BK;                                     // even (K % 32 == 0)
  G32R_LD(va, vb, 0)
  G32R_LD(wa, wb, BK)
  G32R_ST(va, vb, 0)
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    { const int k2 = (kt + 2 < nk ? kt + 2 : nk - 1) * BK; G32R_LD(va, vb, k2) }       // slab kt + 2 (va / vb are free: slab kt is in LDS)
    G32R_SLAB(0)
    G32R_ST(wa, wb, 1)                                                                   // slab kt + 1 -> the other buffer
    __syncthreads();
    { const int k3 = (kt + 3 < nk ? kt + 3 : nk - 1) * BK; G32R_LD(wa, wb, k3) }
    G32R_SLAB(1)
    G32R_ST(va, vb, 0)
    __syncthreads();
  }
#undef G32R_LD
#undef G32R_ST
#undef G32R_SLAB
  // accumulators -> sC[BM][BN + 4] (transposed product: lane (r32, h) holds ONE row per 32x32 tile, registers 4 q .. 4 q + 3 = the
  // four consecutive columns 8 q + 4 h + 0..3), then the shared row epilogue
  constexpr int CSTR = BN + 4;
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      *reinterpret_cast<f32x4*>(&smem[(wm * 32 + r32) * CSTR + wn * (32 * NB) + 32 * t + 8 * q4 + 4 * h]) =
          f32x4{acc[t][4 * q4], acc[t][4 * q4 + 1], acc[t][4 * q4 + 2], acc[t][4 * q4 + 3]};
  __syncthreads();
  gemm_row_epilogue<BM, BN, NT, EPI>(g, m0, by, smem);
}
// host side: does this row-fused Linear qualify, and which instance
static inline bool gemm32row_ok(const GemmArgs& g, bool bkm) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (g.bf16 || (g.N != 256 && g.N != 512) || g.K % 32 || g.K < 32 || g.M % 64) return false;
  if ((g.lda & 3) || (g.ldb & 3) || !al16(g.A) || !al16(g.B)) return false;
  (void)bkm;
  return true;
}
template <bool BKM, int EPI>
static inline void gemm32row_launch(const GemmArgs& g, hipStream_t s) {
  gt_prof_tag(gemm_label<BKM, EPI>(), 2.0 * g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + 3.0 * g.M * g.N));
  // 64-row tiles once they fill the chip (one workgroup per CU at d_model 512: 132 KB of LDS); 32-row tiles below that at 256
  if (g.N == 512) gt_launch(gemm32row_kernel<512, 2, BKM, EPI>, dim3(g.M / 64), dim3(512), s, g);
  else if (g.M / 64 >= GT_ROW32_BM64_MIN_WG) gt_launch(gemm32row_kernel<256, 2, BKM, EPI>, dim3(g.M / 64), dim3(512), s, g);
  else gt_launch(gemm32row_kernel<256, 1, BKM, EPI>, dim3(g.M / 32), dim3(512), s, g);
}

template <bool BKM, int EPI, int PREC = 0>
__global__ __launch_bounds__(256, 2) void gemm32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[Gemm32Cfg::smem<false, BKM>()];
  // XCD-contiguous tile order (tiles that share an A row panel share an L2); placement never changes results
  const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const int xcd = lin & 7, q = nb >> 3, rr = nb & 7;
  const int bid = xcd * q + (xcd < rr ? xcd : rr) + (lin >> 3);
  gemm32_body<false, BKM, EPI, PREC>(g, (bid / gx) * 128, (bid % gx) * 128, 0, g.K / 32, false, smem);
}

template <bool BKM, int EPI>
static inline void gemm32_launch(const GemmArgs& g, hipStream_t s) {
  gt_prof_tag((g.as_dgrad && EPI == EPI_STORE) ? "gemm_dgrad" : gemm_label<BKM, EPI>(), 2.0 * g.M * g.N * g.K,
              4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
  if (g.bf16) gt_launch(gemm32_kernel<BKM, EPI, 1>, dim3(g.N / 128, g.M / 128), dim3(256), s, g);
  else        gt_launch(gemm32_kernel<BKM, EPI, 0>, dim3(g.N / 128, g.M / 128), dim3(256), s, g);
}

// Weight gradients ("TN": dW (out x in) += dY^T X over a token chunk, fp32 atomics; bias gradient = column sums of the dY slabs
// by the tiles of the first column): the grouped dispatch of gt_gemm.h on the big-tile body.  A problem qualifies when its
// out / in sizes are multiples of 128 and its token count of 64 (wgrad32_ok); k_chunk is a multiple of 64.
static inline bool wgrad32_ok(const GemmArgs& g) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return g.M % 128 == 0 && g.N % 128 == 0 && g.K % 64 == 0 && g.K >= 64 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 &&
         al16(g.A) && al16(g.B);
}
// ---- weight gradients with BOTH operands as bf16 shadows (precision = 1, class 4 of the grouped dispatch): the fp32-image body above
// assembles a token-major fragment from eight 4-byte LDS reads per lane -- at the bf16 MFMA rate the LDS read issue, not the matrix
// pipe and not the operand fetch, bounds it (halving the fetched bytes changed nothing: 916 vs 895 us at C5 bs 512).  Here the slabs
// stay bf16 and token-major in LDS ([32 tokens][128 + 32 columns]: 320-byte rows put the four token rows of a transposed read on
// disjoint bank groups) and a fragment is TWO ds_read_b64_tr_b16 -- the hardware transpose read: per 16 lanes a 4-token x 16-column
// block, delivered column-major (cdna_hip_programming.md T10) -- which is exactly the 32x32x16 operand map: lane (r32, h), element j =
// token 16 s + 8 h + j.  Same operand values, same instruction, same k order as the fp32-image body: the per-chunk sums are the same
// numbers.  EXEC is all ones at every transposed read (no divergence above them).
#ifndef GT_WG32T_BK
#define GT_WG32T_BK 32
#endif
#ifndef GT_WG32T_STR
#define GT_WG32T_STR 160
#endif
struct Wg32tCfg { static constexpr int BK = GT_WG32T_BK, STR = GT_WG32T_STR, SZ = BK * STR; };            // bf16 elements
#ifdef GT_EMU
struct __attribute__((may_alias, aligned(8))) Wg32tFrag { uint16_t v[4]; };
__device__ __forceinline__ Wg32tFrag wg32t_tr(const uint16_t* p) { const emu::tr16x4 t = emu::lds_tr16(p); Wg32tFrag r; for (int j = 0; j < 4; ++j) r.v[j] = t.v[j]; return r; }
#else
typedef short Wg32tFrag __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Wg32tFrag wg32t_tr(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) Wg32tFrag*)(p));
}
#endif
__device__ __forceinline__ bf16x8 wg32t_join(const Wg32tFrag& lo, const Wg32tFrag& hi) {
  bf16x8 r;
#ifdef GT_EMU
  for (int j = 0; j < 4; ++j) { r.v[j] = lo.v[j]; r.v[4 + j] = hi.v[j]; }
#else
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 t = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  r = __builtin_bit_cast(bf16x8, t);
#endif
  return r;
}
// sum of the four bf16 values of a transposed-read fragment (the bias gradient = column sums of the dY slabs, taken from the fragments the
// MFMAs consume instead of 32 dependent 2-byte LDS reads per thread and slab in the first-column tiles; measured neutral on the step -- C5 bf16
// bs 64 0.890 ms either way -- kept because it is the shorter code)
__device__ __forceinline__ float wg32t_sum(const Wg32tFrag& f) {
#ifdef GT_EMU
  return (gt_bf2f(f.v[0]) + gt_bf2f(f.v[1])) + (gt_bf2f(f.v[2]) + gt_bf2f(f.v[3]));
#else
  return (gt_bf2f((uint16_t)f[0]) + gt_bf2f((uint16_t)f[1])) + (gt_bf2f((uint16_t)f[2]) + gt_bf2f((uint16_t)f[3]));
#endif
}
__device__ __forceinline__ void wgrad32t_body(const GemmArgs& g, const int m0, const int n0, const int kbeg, const int nk, const bool want_dbias,
                                              uint16_t* sm) {
  typedef Wg32tCfg Cfg;
  constexpr int BK = Cfg::BK, STR = Cfg::STR, SZ = Cfg::SZ, PER = BK / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  G32hRegs a0[PER], b0[PER], a1[PER], b1[PER], a2[PER], b2[PER];
  const uint16_t* pa[PER];
  const uint16_t* pb[PER];
  int so[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, kr = ch >> 4, c8 = (ch & 15) * 8;        // (token, 8 columns) = 16 bytes
    pa[i] = g.A16 + (size_t)(kbeg + kr) * g.lda16 + m0 + c8;
    pb[i] = g.B16 + (size_t)(kbeg + kr) * g.ldb16 + n0 + c8;
    so[i] = kr * STR + c8;
  }
  const size_t astep = (size_t)g.lda16 * BK, bstep = (size_t)g.ldb16 * BK;       // elements per slab
#define W32T_LD(XA, XB, t)                                                                     \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    XA[i] = *reinterpret_cast<const G32hRegs*>(pa[i] + (size_t)(t) * astep);                   \
    XB[i] = *reinterpret_cast<const G32hRegs*>(pb[i] + (size_t)(t) * bstep);                   \
  }
#define W32T_ST(XA, XB, buf)                                                                   \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    *reinterpret_cast<G32hRegs*>(&sm[(buf) * 2 * SZ + so[i]]) = XA[i];                         \
    *reinterpret_cast<G32hRegs*>(&sm[(buf) * 2 * SZ + SZ + so[i]]) = XB[i];                    \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // transposed read of this lane: group gq = lane >> 4 -> columns + 16 (gq & 1), tokens + 8 (gq >> 1) = + 8 h; inside the group lane
  // 4 q + p addresses token row q, columns 4 p ..
  const int L16 = lane & 15, tq = L16 >> 2, tp = L16 & 3;
  const int toff = (8 * h + tq) * STR + 16 * ((lane >> 4) & 1) + 4 * tp;
  const int fa = toff + wm * 64, fb = SZ + toff + wn * 64;
  float bsum[2] = {0.f, 0.f};                      // bias gradient: lane (r32, h) sums row wm * 64 + 32 ti + r32 of dY^T over its tokens (k = 16 s + 8 h + j)
  const bool bias_wave = want_dbias && wn == 0;    // (both column waves hold the same A fragments: one of them sums)
  auto tof = [&](const int t) { return t < nk ? t : nk - 1; };
  W32T_LD(a0, b0, 0)
  W32T_LD(a1, b1, tof(1))
  W32T_LD(a2, b2, tof(2))
  W32T_ST(a0, b0, 0)
  W32T_LD(a0, b0, tof(3))
  __syncthreads();
#define W32T_SLAB(CUR, NA, NB, t)                                                              \
  if ((t) < nk) {                                                                              \
  _Pragma("unroll") for (int s_ = 0; s_ < BK / 16; ++s_) {                                     \
    bf16x8 a16[2], b16[2];                                                                     \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) {                                         \
      const uint16_t* qa = &sm[(CUR) * 2 * SZ + fa + 32 * ti + 16 * s_ * STR];                 \
      const uint16_t* qb = &sm[(CUR) * 2 * SZ + fb + 32 * ti + 16 * s_ * STR];                 \
      const Wg32tFrag alo_ = wg32t_tr(qa), ahi_ = wg32t_tr(qa + 4 * STR);                      \
      a16[ti] = wg32t_join(alo_, ahi_);                                                        \
      b16[ti] = wg32t_join(wg32t_tr(qb), wg32t_tr(qb + 4 * STR));                              \
      if (bias_wave) bsum[ti] += wg32t_sum(alo_) + wg32t_sum(ahi_);   /* (wave-uniform) */     \
    }                                                                                          \
    _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                           \
    _Pragma("unroll") for (int tb = 0; tb < 2; ++tb)                                           \
      acc[ta][tb] = GT_MFMA32_BF16(a16[ta], b16[tb], acc[ta][tb]);                             \
  }                                                                                            \
  W32T_ST(NA, NB, (CUR) ^ 1)                                                                   \
  W32T_LD(NA, NB, tof((t) + 4))                                                                \
  __syncthreads();                                                                             \
  }
  for (int kt = 0; kt < nk; kt += 6) {
    W32T_SLAB(0, a1, b1, kt) W32T_SLAB(1, a2, b2, kt + 1) W32T_SLAB(0, a0, b0, kt + 2)
    W32T_SLAB(1, a1, b1, kt + 3) W32T_SLAB(0, a2, b2, kt + 4) W32T_SLAB(1, a0, b0, kt + 5)
  }
#undef W32T_SLAB
#undef W32T_LD
#undef W32T_ST
  if (bias_wave) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const float t = bsum[ti] + __shfl_xor(bsum[ti], 32);
      if (h == 0) atomicAdd(&g.dbias[m0 + wm * 64 + 32 * ti + r32], t);
    }
  }
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + ta * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn * 64 + tb * 32 + r32;
        atomicAdd(&g.C[(size_t)row * g.ldc + col], acc[ta][tb][e]);
      }
}
#ifndef GT_WG32T_OCC
#define GT_WG32T_OCC 2
#endif
__global__ __launch_bounds__(256, GT_WG32T_OCC) void wgrad32t_group_kernel(GemmGroup grp) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[4 * Wg32tCfg::SZ];        // [buffer][A | B]
  const int nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, r = nb & 7;
  const int b = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
  int i = 0;
  while (i + 1 < grp.n && b >= grp.start[i + 1]) ++i;
  const GemmArgs& g = grp.p[i];
  const int local = b - grp.start[i];
  const int bx = local % grp.gx[i], t = local / grp.gx[i], by = t % grp.gy[i], bz = t / grp.gy[i];
  const int kbeg = bz * g.k_chunk;
  const int kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;
  wgrad32t_body(g, by * 128, bx * 128, kbeg, (kend - kbeg) / Wg32tCfg::BK, g.dbias != nullptr && bx == 0, sm);
}

template <int PREC, bool SA16, bool SB16>
__global__ __launch_bounds__(256, 2) void wgrad32_group_kernel(GemmGroup grp) {
  __shared__ __attribute__((aligned(16))) float smem[Gemm32Cfg::smem<true, true>()];
  const int nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, r = nb & 7;
  const int b = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
  int i = 0;
  while (i + 1 < grp.n && b >= grp.start[i + 1]) ++i;
  const GemmArgs& g = grp.p[i];
  const int local = b - grp.start[i];
  const int bx = local % grp.gx[i], t = local / grp.gx[i], by = t % grp.gy[i], bz = t / grp.gy[i];
  const int kbeg = bz * g.k_chunk;
  const int kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;
  gemm32_body<true, true, EPI_ATOMIC, PREC, SA16, SB16>(g, by * 128, bx * 128, kbeg, (kend - kbeg) / 32, g.dbias != nullptr && bx == 0, smem);
}
