// Mid-size problems (round 5): 64x64 output tiles on the prefetch-ring body -- the Linears / dgrads of d_model 512 at a GPU's share of a
// data-parallel batch (2048 tokens: N = 512 gives 64 tiles of 128x128 for 256 CUs, and even the QKV projection only 192), and of d_model
// 256 at 8192 tokens.  Until round 5 those ran on 32x32 tiles of the one-deep 16x16x4 body (1024 workgroups, 52-53 % of the fp32 MFMA
// peak at d_model 512 / 2048 tokens) or on 192 tiles of 128x128 (one round on 192 of 256 CUs: 27 us of MFMA issue per tile against 20.5 us
// for the same flops spread over the chip).
//
// Shape: 256 threads = 2 x 2 waves, each wave owns ONE 32x32 block of the tile over the whole contraction (v_mfma_f32_32x32x2_f32; ONE
// accumulation chain: a 16-pass MFMA issues back to back on its own result, MI355X_MICROARCH.md cycle constants); K in 64-wide slabs, double-buffered in LDS in SOURCE orientation
// ([row][64 + 4] for k-contiguous operands: a fragment = one ds_read_b128; [k][64 + 4] for the dgrad's weight: four ds_read_b32), slab
// t + 2 in flight in registers while slab t + 1 waits in registers and slab t computes (the two-deep ring of gt_gemm32.h), ONE staging
// instruction behind each MFMA (sched_group_barrier).  A slab = 8 groups of 4 MFMAs per wave: 8 global loads + 8 LDS writes + 16 fragment
// reads for 32 MFMAs.  70 KB of LDS: two workgroups per CU where the grid has them (QKV at 2048 tokens: 768 tiles = 3 per CU).
// The store epilogues are those of gt_gemm32.h (same lane -> element map, TA = TB = 1).
//
// k order inside a slab: group kk = 0..7 (8 k each), MFMA j = 0..3 contracts k = 8 kk + {j, 4 + j}.  Any fp32 fmaf chain
// over k in another order gives the same numbers up to the order of the additions (tests compare with the oracle, not bit for bit with
// the other tile classes).
#pragma once
// (included at the end of gt_gemm.h, after gt_gemm32.h)

struct Gemm64Cfg {
  static constexpr int BM = 64, BN = 64, BK = 64, NT = 256;
  static constexpr int STR = 64 + 4, SZ = 64 * STR;           // floats: [64 rows][BK + 4] or [BK k][64 + 4]
  static constexpr int SMEM = 4 * SZ;                          // [A buf 0 | A buf 1 | B buf 0 | B buf 1]
};

// interior tiles only, an even number of 64-wide slabs, 16-byte rows
static inline bool gemm64_ok(const GemmArgs& g, int epi) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (g.M % 64 || g.N % 64 || g.K % 128 || g.K < 128) return false;
  if ((g.lda & 3) || (g.ldb & 3) || (g.ldc & 3) || !al16(g.A) || !al16(g.B) || (g.C && !al16(g.C))) return false;
  if (epi == EPI_STORE || epi == EPI_RELU_DROP) { if (g.bias && !al16(g.bias)) return false; }
  if ((epi == EPI_MASK_NZ || epi == EPI_ADD_RELUMASK_DROP) && ((g.ldres & 3) || !al16(g.res) || (g.res16 && (reinterpret_cast<uintptr_t>(g.res16) & 7)))) return false;
  if (epi == EPI_ADD_RELUMASK_DROP && ((g.N & 3) || !al16(g.aux_in))) return false;
  if (g.C16 && ((g.ldc16 & 3) || (reinterpret_cast<uintptr_t>(g.C16) & 7))) return false;
  return true;
}

// XCD-contiguous tile order (tiles that share an A row panel share an L2); placement never changes results
__device__ __forceinline__ int gemm64_bid() {
  const int gx = gridDim.x, nb = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const int xcd = lin & 7, q = nb >> 3, rr = nb & 7;
  return xcd * q + (xcd < rr ? xcd : rr) + (lin >> 3);
}

// staging instructions behind the 4 MFMAs of one group: NX of class MASKX first (one per MFMA), the NR fragment reads from the front
#define G64_M GT_SGB(0x8, 1)
#define G64_X(mask) GT_SGB(mask, 1)
#define G64_R(n) GT_SGB(0x100, n)

template <bool BKM, int EPI, int PREC = 0>
__global__ __launch_bounds__(256, 2) void gemm64_kernel(GemmArgs g) {
  typedef Gemm64Cfg Cfg;
  constexpr int BK = Cfg::BK, STR = Cfg::STR, SZ = Cfg::SZ, PER = 4;
  __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM];
  const int bid = gemm64_bid(), gx = gridDim.x;
  const int m0 = (bid / gx) * 64, n0 = (bid % gx) * 64, nk = g.K / BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;

  f32x4 va[PER], vb[PER], wa[PER], wb[PER];
  const char* pa[PER];
  const char* pb[PER];
  int so_a[PER], so_b[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch >> 4, c = (ch & 15) * 4;         // (row, 4 k) -- or, BKM, (k, 4 columns)
    pa[i] = reinterpret_cast<const char*>(g.A + (size_t)(m0 + r) * g.lda + c);
    so_a[i] = r * STR + c;
    pb[i] = BKM ? reinterpret_cast<const char*>(g.B + (size_t)r * g.ldb + n0 + c) : reinterpret_cast<const char*>(g.B + (size_t)(n0 + r) * g.ldb + c);
    so_b[i] = r * STR + c;
  }
  const size_t bstep = BKM ? (size_t)g.ldb * 4 : 4;                        // bytes per k
#define G64_LD(XA, XB, k0)                                                                     \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    XA[i] = *reinterpret_cast<const f32x4*>(pa[i] + (size_t)(k0) * 4);                         \
    XB[i] = *reinterpret_cast<const f32x4*>(pb[i] + (size_t)(k0) * bstep);                     \
  }
#define G64_ST(XA, XB, buf)                                                                    \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    *reinterpret_cast<f32x4*>(&smem[(buf) * SZ + so_a[i]]) = XA[i];                            \
    *reinterpret_cast<f32x4*>(&smem[2 * SZ + (buf) * SZ + so_b[i]]) = XB[i];                   \
  }
  f32x16 acc[1][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
  const int offa = (wm * 32 + r32) * STR + 4 * h;
  const int offb = 2 * SZ + (BKM ? (4 * h) * STR + wn * 32 + r32 : (wn * 32 + r32) * STR + 4 * h);

  G64_LD(va, vb, 0)
  G64_LD(wa, wb, BK)
  G64_ST(va, vb, 0)
  __syncthreads();

  if constexpr (PREC == 1) {
    // bf16 operands from fp32 sources: rounded when a lane assembles its fragment -- k-step s_ (16 k): lane half h takes k = 16 s_ + 8 h + j
    // (the map of gt_gemm32.h's PREC = 1 body).  16x fewer matrix cycles: bound by the staging, plain one-barrier-per-slab schedule.
    auto frag_a = [&](const int buf, const int s_) -> bf16x8 {
      bf16x8 r;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + buf * SZ + (wm * 32 + r32) * STR + 16 * s_ + 8 * h);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + buf * SZ + (wm * 32 + r32) * STR + 16 * s_ + 8 * h + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { GT_BF16X8_SET(r, j, lo[j]); GT_BF16X8_SET(r, 4 + j, hi[j]); }
      return r;
    };
    auto frag_b = [&](const int buf, const int s_) -> bf16x8 {
      bf16x8 r;
      if (BKM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) GT_BF16X8_SET(r, j, smem[2 * SZ + buf * SZ + (16 * s_ + 8 * h + j) * STR + wn * 32 + r32]);
      } else {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + 2 * SZ + buf * SZ + (wn * 32 + r32) * STR + 16 * s_ + 8 * h);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + 2 * SZ + buf * SZ + (wn * 32 + r32) * STR + 16 * s_ + 8 * h + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { GT_BF16X8_SET(r, j, lo[j]); GT_BF16X8_SET(r, 4 + j, hi[j]); }
      }
      return r;
    };
#define G64_SLAB16(CUR, NA, NB, FA_, FB_, t)                                                   \
    { const int k2_ = ((t) + 2 < nk ? (t) + 2 : nk - 1) * BK;                                  \
      G64_LD(FA_, FB_, k2_) }                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                         \
      const bf16x8 a16 = frag_a(CUR, s_), b16 = frag_b(CUR, s_);                               \
      acc[0][0] = GT_MFMA32_BF16(b16, a16, acc[0][0]);                                         \
    }                                                                                          \
    G64_ST(NA, NB, (CUR) ^ 1)                                                                  \
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      G64_SLAB16(0, wa, wb, va, vb, kt)
      G64_SLAB16(1, va, vb, wa, wb, kt + 1)
    }
#undef G64_SLAB16
  } else {
    // fragments of one 8-k group: element j of lane half h is k = 4 h + j of the group, for A and B alike
    f32x4 fa0, fb0, fa1, fb1;
#define G64_RD(FA, FB, buf, kk)                                                                \
    FA = *reinterpret_cast<const f32x4*>(smem + (buf) * SZ + offa + (kk) * 8);                 \
    if (BKM) {                                                                                 \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) FB[j] = smem[(buf) * SZ + offb + ((kk) * 8 + j) * STR]; \
    } else {                                                                                   \
      FB = *reinterpret_cast<const f32x4*>(smem + (buf) * SZ + offb + (kk) * 8);               \
    }
    // (transposed product, as the store epilogue expects: first operand = the B fragment)
#define G64_MM(FA, FB)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[0][0] = GT_MFMA32(FB[j], FA[j], acc[0][0]);
    constexpr int NR = BKM ? 5 : 2;                            // LDS reads per fragment pair
    // the reads of a group spread over its 4 MFMAs from the front: 2 -> 1 1 0 0; 5 -> 2 1 1 1
#define G64_RS0 if constexpr (BKM) { G64_R(2) } else { G64_R(1) }
#define G64_RS1 G64_R(1)
#define G64_RS2 if constexpr (BKM) { G64_R(1) }
#define G64_RS3 if constexpr (BKM) { G64_R(1) }
    // one group: 4 MFMAs, NX (<= 3) instructions of class MASKX, the next group's fragment reads
#define G64_GRP3(mask) G64_M G64_X(mask) G64_RS0 G64_M G64_X(mask) G64_RS1 G64_M G64_X(mask) G64_RS2 G64_M G64_RS3
#define G64_GRP2(mask) G64_M G64_X(mask) G64_RS0 G64_M G64_X(mask) G64_RS1 G64_M G64_RS2 G64_M G64_RS3
#define G64_GRP0() G64_M G64_RS0 G64_M G64_RS1 G64_M G64_RS2 G64_M G64_RS3
    G64_RD(fa0, fb0, 0, 0)
    // one slab.  CUR: LDS buffer holding slab t; (NA, NB): registers holding slab t + 1; (FA_, FB_): the set slab t came from, free
    // again -> receives slab t + 2 (the last two slabs re-load the final slab: branch-free, never used)
    // (LDS writes and reads may alias as far as the compiler knows, so it keeps their program order: the eight writes of slab t + 1 are
    //  cut into three runs, each ahead of the fragment reads of its group)
#define G64_STA(XA, buf, i) *reinterpret_cast<f32x4*>(&smem[(buf) * SZ + so_a[i]]) = XA[i];
#define G64_STB(XB, buf, i) *reinterpret_cast<f32x4*>(&smem[2 * SZ + (buf) * SZ + so_b[i]]) = XB[i];
    // a group with writes: (M W)(M W)(M W R..)(M R..) / (M W)(M W)(M R..)(M R..)
#define G64_RSA if constexpr (BKM) { G64_R(2) } else { G64_R(1) }
#define G64_RSB if constexpr (BKM) { G64_R(3) } else { G64_R(1) }
#define G64_GRPW3() G64_M G64_X(0x200) G64_M G64_X(0x200) G64_M G64_X(0x200) G64_RSA G64_M G64_RSB
#define G64_GRPW2() G64_M G64_X(0x200) G64_M G64_X(0x200) G64_M G64_RSA G64_M G64_RSB
#define G64_SLAB(CUR, NA, NB, FA_, FB_, t)                                                     \
    { const int k2_ = ((t) + 2 < nk ? (t) + 2 : nk - 1) * BK;                                  \
      G64_LD(FA_, FB_, k2_) }                                                                  \
    G64_RD(fa1, fb1, CUR, 1) G64_MM(fa0, fb0)                                                  \
    G64_RD(fa0, fb0, CUR, 2) G64_MM(fa1, fb1)                                                  \
    G64_RD(fa1, fb1, CUR, 3) G64_MM(fa0, fb0)                                                  \
    G64_RD(fa0, fb0, CUR, 4) G64_MM(fa1, fb1)                                                  \
    G64_GRP3(0x20) G64_GRP3(0x20) G64_GRP2(0x20) G64_GRP0() GT_SCHED_FENCE()                   \
    G64_STA(NA, (CUR) ^ 1, 0) G64_STB(NB, (CUR) ^ 1, 0) G64_STA(NA, (CUR) ^ 1, 1)              \
    G64_RD(fa1, fb1, CUR, 5) G64_MM(fa0, fb0)                                                  \
    G64_STB(NB, (CUR) ^ 1, 1) G64_STA(NA, (CUR) ^ 1, 2) G64_STB(NB, (CUR) ^ 1, 2)              \
    G64_RD(fa0, fb0, CUR, 6) G64_MM(fa1, fb1)                                                  \
    G64_STA(NA, (CUR) ^ 1, 3) G64_STB(NB, (CUR) ^ 1, 3)                                        \
    G64_RD(fa1, fb1, CUR, 7) G64_MM(fa0, fb0)                                                  \
    G64_GRPW3() G64_GRPW3() G64_GRPW2() GT_SCHED_FENCE()                                       \
    __syncthreads();                                                                           \
    G64_RD(fa0, fb0, (CUR) ^ 1, 0) G64_MM(fa1, fb1) G64_GRP0() GT_SCHED_FENCE()
    for (int kt = 0; kt < nk; kt += 2) {
      G64_SLAB(0, wa, wb, va, vb, kt)
      G64_SLAB(1, va, vb, wa, wb, kt + 1)
    }
#undef G64_SLAB
#undef G64_RD
#undef G64_MM
  }
#undef G64_LD
#undef G64_ST
  gemm32_store_epilogue<EPI, 1, 1>(g, acc, m0, n0, wm, wn, r32, h);
}

// GT_TRACE_GEMM64=1: one line on stderr per launch (tests assert that a shape really took this kernel)
static inline void gemm64_trace(const char* what, const GemmArgs& g, bool bkm, int epi) {
  static const int on = [] { const char* e = getenv("GT_TRACE_GEMM64"); return (e && e[0] == '1') ? 1 : 0; }();
  if (on) fprintf(stderr, "[gemm64] %s M %d N %d K %d %s epi %d prec %d\n", what, g.M, g.N, g.K, bkm ? "NN" : "NT", epi, g.bf16);
}
template <bool BKM, int EPI>
static inline void gemm64_launch(const GemmArgs& g, hipStream_t s) {
  gemm64_trace("fp32-source", g, BKM, EPI);
  gt_prof_tag((g.as_dgrad && EPI == EPI_STORE) ? "gemm_dgrad" : gemm_label<BKM, EPI>(), 2.0 * g.M * g.N * g.K,
              4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
  if (g.bf16) gt_launch(gemm64_kernel<BKM, EPI, 1>, dim3(g.N / 64, g.M / 64), dim3(256), s, g);
  else        gt_launch(gemm64_kernel<BKM, EPI, 0>, dim3(g.N / 64, g.M / 64), dim3(256), s, g);
}

// ================================================================================================================ bf16 SOURCES
// precision = 1 with operand shadows (gt_gemm32.h, gemm32h_kernel) on the same 64x64 tile: both operands k-contiguous bf16 (g.A16 [M][K],
// g.B16 [N][K]); 128-wide slabs (16 KB per operand... 64 rows x 256 B), bf16 LDS images [64][128 + 8] (272-byte rows: a fragment = one
// conflict-free ds_read_b128), a three-deep register ring, v_mfma_f32_32x32x16_bf16 with gemm32h's k -> (lane half, element) map.  At
// 2048 tokens the 128x128 form ran 64 tiles on 64 of 256 CUs (12-14 us per Linear at K = 512: a latency chain on a quarter of the chip).
struct Gemm64hCfg {
  static constexpr int BK = 128, STR = BK + 8, SZ = 64 * STR;           // bf16 elements
};
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm64h_kernel(GemmArgs g) {
  typedef Gemm64hCfg Cfg;
  constexpr int BK = Cfg::BK, STR = Cfg::STR, SZ = Cfg::SZ, PER = 4;
  __shared__ __attribute__((aligned(16))) uint16_t sm[4 * SZ];           // [buffer][A | B]
  const int bid = gemm64_bid(), gx = gridDim.x;
  const int m0 = (bid / gx) * 64, n0 = (bid % gx) * 64, nk = g.K / BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  G32hRegs a0[PER], b0[PER], a1[PER], b1[PER], a2[PER], b2[PER];
  const uint16_t* pa[PER];
  const uint16_t* pb[PER];
  int so[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * 256, r = ch >> 4, c = (ch & 15) * 8;        // (row, 8 k) = 16 bytes
    pa[i] = g.A16 + (size_t)(m0 + r) * g.lda16 + c;
    pb[i] = g.B16 + (size_t)(n0 + r) * g.ldb16 + c;
    so[i] = r * STR + c;
  }
#define G64H_LD(XA, XB, k0)                                                                    \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    XA[i] = *reinterpret_cast<const G32hRegs*>(pa[i] + (k0));                                  \
    XB[i] = *reinterpret_cast<const G32hRegs*>(pb[i] + (k0));                                  \
  }
#define G64H_ST(XA, XB, buf)                                                                   \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                            \
    *reinterpret_cast<G32hRegs*>(&sm[(buf) * 2 * SZ + so[i]]) = XA[i];                         \
    *reinterpret_cast<G32hRegs*>(&sm[(buf) * 2 * SZ + SZ + so[i]]) = XB[i];                    \
  }
  f32x16 acc[1][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
  const int fa = (wm * 32 + r32) * STR + 8 * h, fb = SZ + (wn * 32 + r32) * STR + 8 * h;
  auto kof = [&](const int t) { return (t < nk ? t : nk - 1) * BK; };
  G64H_LD(a0, b0, 0)
  G64H_LD(a1, b1, kof(1))
  G64H_LD(a2, b2, kof(2))
  G64H_ST(a0, b0, 0)
  G64H_LD(a0, b0, kof(3))
  __syncthreads();
  // slab t from LDS buffer CUR; (NA, NB) hold slab t + 1: written to the other buffer, then reloaded with slab t + 4
#define G64H_SLAB(CUR, NA, NB, t)                                                              \
  if ((t) < nk) {                                                                              \
  _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                           \
    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(&sm[(CUR) * 2 * SZ + fa + 16 * s_]);    \
    const bf16x8 y0 = *reinterpret_cast<const bf16x8*>(&sm[(CUR) * 2 * SZ + fb + 16 * s_]);    \
    acc[0][0] = GT_MFMA32_BF16(y0, x0, acc[0][0]);                                             \
  }                                                                                            \
  G64H_ST(NA, NB, (CUR) ^ 1)                                                                   \
  G64H_LD(NA, NB, kof((t) + 4))                                                                \
  __syncthreads();                                                                             \
  }
  for (int kt = 0; kt < nk; kt += 6) {
    G64H_SLAB(0, a1, b1, kt) G64H_SLAB(1, a2, b2, kt + 1) G64H_SLAB(0, a0, b0, kt + 2)
    G64H_SLAB(1, a1, b1, kt + 3) G64H_SLAB(0, a2, b2, kt + 4) G64H_SLAB(1, a0, b0, kt + 5)
  }
#undef G64H_SLAB
#undef G64H_LD
#undef G64H_ST
  gemm32_store_epilogue<EPI, 1, 1>(g, acc, m0, n0, wm, wn, r32, h);
}
// host side: shadows present, interior tiles, whole 128-wide slabs, 16-byte rows
static inline bool gemm64h_ok(const GemmArgs& g, int epi) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!g.A16 || !g.B16 || (g.lda16 & 7) || (g.ldb16 & 7) || !al16(g.A16) || !al16(g.B16)) return false;
  if (g.accumulate) return false;
  GemmArgs t = g; t.A = reinterpret_cast<const float*>(g.A16); t.B = reinterpret_cast<const float*>(g.B16); t.lda = t.ldb = 4;
  return gemm64_ok(t, epi);
}
template <bool BKM, int EPI>
static inline void gemm64h_launch(const GemmArgs& g, hipStream_t s) {
  gemm64_trace("bf16-source", g, BKM, EPI);
  gt_prof_tag(gemm_label<BKM, EPI>(), 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + 4.0 * (double)g.M * g.N);
  gt_launch(gemm64h_kernel<EPI>, dim3(g.N / 64, g.M / 64), dim3(256), s, g);
}
