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

#ifndef GT_G64_DEEP
#define GT_G64_DEEP 0           /* fp32 MFMA loop: 1 = a three-deep register ring (slab t + 4 requested in the second half of slab t), 0 = the
                                   two-deep ring of gt_gemm32.h.  Measured (round 5, gemm_bench, M 2048): three-deep 13.6 / 32.5 / 34.6 us against
                                   12.8 / 31.2 / 33.4 us (N 512 K 512 / N 512 K 1536 / N 1536 K 512): bytes in flight are not what holds the loop at
                                   1.15 us per 64-wide slab (MFMA issue 0.85 us at 2.4 GHz -- ~0.97 us at the ~2.1 GHz the chip holds under this load) */
#endif
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
  if (epi == EPI_RES_LN || epi == EPI_RES_LNBWD) {
    if (!g.rowx || !g.gamma || !al16(g.gamma) || (g.res && ((g.ldres & 3) || !al16(g.res))) || !g.C) return false;
    if (epi == EPI_RES_LN && (!g.beta || !al16(g.beta) || !g.aux || !al16(g.aux) || !g.aux2 || (g.bias && !al16(g.bias)))) return false;
    if (epi == EPI_RES_LNBWD && (!g.xhat || !al16(g.xhat) || !g.rstd || !g.ln_part || (g.C2 && !al16(g.C2)))) return false;
  }
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

// ================================================================================================================ LayerNorm epilogues
// EPI_RES_LN / EPI_RES_LNBWD on 64x64 tiles (round 5).  A LayerNorm needs sums over the WHOLE row (N = d_model columns = N / 64 column
// tiles of a row block), and at 2048 tokens a row-owning tile makes every workgroup stream the whole weight matrix (23 TF measured) --
// so until round 5 the norm was a launch of its own behind the Linear / dgrad: 24 launches of 5.8-7.3 us at d_model 512 (9.5 % of the
// step).  Here the N / 64 workgroups of a row block exchange their ROW PARTIALS inside the launch and each normalises its own tile:
//   * every wave owns 32 rows x 32 columns ("part" p = column / 32): per row two partial values -- forward (mean, M2) of its 32 columns
//     (two passes over the registers; parts merged by Chan's formula in a fixed order: every workgroup arrives at bit-identical
//     statistics); backward (sum g, sum g xhat), g = dy gamma;
//   * hand-off = ONE hop: a value travels as an 8-byte granule {value bits, sequence number} written by one agent-scope (write-through, never
//     torn) store and polled by agent-scope loads -- the QUAD pair exchange of gt_seq.h widened to N / 32 parties.  Layout [row block][part]
//     [row half][value][32 rows]: a wave publishes 512 contiguous bytes, and the 32 lanes of a half poll 256 contiguous bytes per load (2 x N / 32
//     loads per lane and round).  Three versions were measured at d_model 512 / 2048 tokens: (1) granules laid out per ROW (16 scattered 8-byte
//     loads per lane and round: ~1 M L2 requests per round chip-wide) 13 us per launch; (2) "data, drain, ready word, poll, fetch" (three hops)
//     5.3 us -- what the LayerNorm pass of its own costs; (3) this one;
//   * sequence number: every granule advances by exactly one per launch, so a lane reads its OWN granule's number at its start and adds one
//     -- no global counter, nothing is ever zeroed, a stale granule can never match (the host emulator re-runs workgroups: there the
//     launch serial of the emulator);
//   * the workgroups of a row block must be resident together: the host takes this path only when the whole grid fits the chip at once
//     (gemm64_ln_shape), they are 8 consecutive tiles of one XCD, and the polling loop is bounded -- a time-out raises the header's error
//     word and the update kernels then apply nothing (groove_hip.h, gt_set_xchg_spin_max).
// The arithmetic per element is that of ln_fwd_kernel / ln_bwd_v4_kernel (gt_misc.h).
// (The host emulator runs a workgroup's lanes as fibers, one after the other, and re-runs a workgroup whose poll fails from its start: every
//  lane must have published before the first one polls, and the decision to re-run must be the whole workgroup's.)
#ifdef GT_EMU
#define G64_EMU_PUBLISHED() __syncthreads()
static bool g64_emu_fail_ = false;
#define G64_EMU_AGREE(ok) { __syncthreads(); g64_emu_fail_ = false; __syncthreads(); if (!(ok)) g64_emu_fail_ = true; __syncthreads(); if (g64_emu_fail_) emu::block_retry(); }
#else
#define G64_EMU_PUBLISHED()
#define G64_EMU_AGREE(ok) (void)(ok);
#endif
#define GT_ROWX_HDR 64                                   /* floats: [0] error word */
#ifndef GT_ROWX_SPIN_MAX
#define GT_ROWX_SPIN_MAX (1 << 18)                       /* polls before a workgroup gives its row block up (~0.2 s; a partner delayed by another stream's kernel arrives within milliseconds) */
#endif
// region: header | granules [M / 64 row blocks][N / 32 parts][2 row halves][2 values][32 rows] of 8 bytes
static inline int64_t gt_rowx_floats(int64_t M, int N) { return GT_ROWX_HDR + (M / 64) * (int64_t)(N / 32) * 128 * 2; }
__device__ __forceinline__ unsigned long long g64_ld(const unsigned long long* p) {
#ifdef GT_EMU
  return *p;
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// this launch's sequence number, from the lane's OWN granule
__device__ __forceinline__ uint32_t g64_seq(const unsigned long long* mine) {
#ifdef GT_EMU
  (void)mine;
  return emu::launch_serial;
#else
  return (uint32_t)(g64_ld(mine) >> 32) + 1u;
#endif
}
__device__ __forceinline__ void g64_publish(unsigned long long* mine, const float v, const uint32_t seq) {
  const unsigned long long w = ((unsigned long long)seq << 32) | (unsigned long long)gt_f2u(v);
#ifdef GT_EMU
  *mine = w;
#else
  __hip_atomic_store(mine, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// the NG = 2 x (NP / 2) granules this lane combines -- parts [h NP / 2, (h + 1) NP / 2) of its row, (value 0, value 1) each; `base` = the row
// block's granules + this lane's (row half, row) offset.  false: gave up (error word raised) / emulator: a partner has not run yet
template <int NG>
__device__ __forceinline__ bool g64_collect(const unsigned long long* base, const int h, const uint32_t seq, float (&v)[NG], unsigned* err, const int spin_max) {
  unsigned long long w[NG];
  bool ok = true;
  auto load_all = [&]() {
    ok = true;
#pragma unroll
    for (int i = 0; i < NG / 2; ++i) {
      const unsigned long long* src = base + (size_t)(h * (NG / 2) + i) * 128;       // part stride: [2 row halves][2 values][32 rows]
      w[2 * i] = g64_ld(src); w[2 * i + 1] = g64_ld(src + 32);
      ok = ok && (uint32_t)(w[2 * i] >> 32) == seq && (uint32_t)(w[2 * i + 1] >> 32) == seq;
    }
  };
#ifdef GT_EMU
  load_all();                                                    // (missing: G64_EMU_AGREE re-runs the workgroup after the others)
#else
  // (a word raised by an EARLIER launch -- the host has not fallen back yet -- means this launch's partners may be as absent as that one's:
  //  one look, no spin; every update is skipped anyway until the host has seen the word)
  int spins = 0;
  for (;;) {
    load_all();
    if (__all(ok)) break;                                        // (wave-uniform exit)
    if (spins == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spins = spin_max;       // (looked at only once the first poll failed)
    if (++spins > spin_max) { if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    __builtin_amdgcn_s_sleep(2);
  }
#endif
#pragma unroll
  for (int j = 0; j < NG; ++j) v[j] = gt_u2f((uint32_t)w[j]);
  return ok;
}
// NPH = N / 64 column tiles per row block (4: d_model 256, 8: d_model 512).  acc: the wave's 32x32 block in the store epilogue's lane map
// (lane (r32, h): ONE row, registers 4 q + j = column 8 q + 4 h + j of the block).  smem: the operand buffers, free behind the main loop.
// granules of row block m0 / 64: [part][row half][value][32 rows]
__device__ __forceinline__ unsigned long long* g64_rowx(const GemmArgs& g, const int m0) {
  return reinterpret_cast<unsigned long long*>(g.rowx + GT_ROWX_HDR) + (size_t)(m0 >> 6) * (g.N / 32) * 128;
}
template <int EPI, int NPH>
__device__ __forceinline__ void gemm64_ln_epilogue(const GemmArgs& g, const f32x16& acc, const int m0, const int n0, const int wm, const int wn,
                                                   const int r32, const int h, const uint32_t seq, float* smem) {
  constexpr int N = 64 * NPH, NP = 2 * NPH, NG = NP;          // parts of 32 columns per row; granules a lane combines (NP / 2 parts x 2 values)
  const int row = m0 + wm * 32 + r32, cb = n0 + wn * 32 + 4 * h;       // this lane's row; its columns: cb + 8 q + j
  unsigned* const xerr = g.rowx;
  unsigned long long* const xrb = g64_rowx(g, m0);
  const int part = (n0 >> 5) + wn;
  unsigned long long* const mine = xrb + (size_t)part * 128 + wm * 64 + h * 32 + r32;       // this lane's granule: value h of its row
  const unsigned long long* const theirs = xrb + wm * 64 + r32;                               // + part * 128 (+ 32: value 1)
  const uint32_t dkey = gt_drop_key(g.drop);
  const float invN = 1.0f / (float)N;
  f32x4 ga[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ga[q] = *reinterpret_cast<const f32x4*>(g.gamma + cb + 8 * q);
  if constexpr (EPI == EPI_RES_LN) {
    // z = drop(acc + bias) + res;  y = LN(z) gamma + beta;  aux = xhat, aux2 = rstd, C16 = bf16(y)
    f32x4 bi[4], re[4], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bi[q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + cb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      re[q] = g.res ? *reinterpret_cast<const f32x4*>(g.res + (size_t)row * g.ldres + cb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      be[q] = *reinterpret_cast<const f32x4*>(g.beta + cb + 8 * q);
    }
    float z[16], s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lin = acc[4 * q + j] + bi[q][j];
        if (g.round16) lin = gt_bf2f(gt_f2bf(lin));                // (precision 2: the value the un-fused form stores in bf16 ahead of its norm)
        const float v = lin * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + cb + 8 * q + j)) + re[q][j];
        z[4 * q + j] = v; s += v;
      }
    s += __shfl_xor(s, 32);
    const float mw = s * (1.0f / 32.0f);
    float qq = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = z[e] - mw; qq += d * d; }
    qq += __shfl_xor(qq, 32);
    g64_publish(mine, h ? qq : mw, seq);                        // (lane half 0: the means, half 1: M2)
    G64_EMU_PUBLISHED();
    float pv[NG];
    const bool gotf = g64_collect<NG>(theirs, h, seq, pv, xerr, g.spin_max);       // (mean, M2) pairs of this half's parts
    G64_EMU_AGREE(gotf)
    // Chan's merge over this half's parts, in order (32 columns each) ...
    float mean = pv[0], m2 = pv[1], cnt = 32.f;
#pragma unroll
    for (int i = 1; i < NG / 2; ++i) {
      const float d = pv[2 * i] - mean, n = cnt + 32.f;
      mean += d * (32.f / n);
      m2 += pv[2 * i + 1] + d * d * (cnt * 32.f / n);
      cnt = n;
    }
    // ... then the two halves, lower columns first on both lanes: identical statistics in every lane and workgroup of the row
    const float om = __shfl_xor(mean, 32), o2 = __shfl_xor(m2, 32);
    const float ma = h ? om : mean, a2 = h ? o2 : m2, mb = h ? mean : om, b2 = h ? m2 : o2;
    const float d = mb - ma;
    mean = ma + d * 0.5f;
    m2 = a2 + b2 + d * d * (cnt * 0.5f);
    const float rstd = 1.0f / sqrtf(m2 * invN + GT_LN_EPS);
    if (n0 == 0 && wn == 0 && h == 0) g.aux2[row] = rstd;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 xh, yv;
#pragma unroll
      for (int j = 0; j < 4; ++j) { xh[j] = (z[4 * q + j] - mean) * rstd; yv[j] = xh[j] * ga[q][j] + be[q][j]; }
      *reinterpret_cast<f32x4*>(g.aux + (size_t)row * N + cb + 8 * q) = xh;
      *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + cb + 8 * q) = yv;
      if (g.C16 != nullptr) {
        uint2 pk;
        pk.x = (uint32_t)gt_f2bf(yv[0]) | ((uint32_t)gt_f2bf(yv[1]) << 16); pk.y = (uint32_t)gt_f2bf(yv[2]) | ((uint32_t)gt_f2bf(yv[3]) << 16);
        *reinterpret_cast<uint2*>(g.C16 + (size_t)row * g.ldc16 + cb + 8 * q) = pk;
      }
    }
  } else {
    // d = acc (+ res);  dz = LNbwd(d) with (xhat, rstd, gamma);  C2 = dz dropmask;  C16 = bf16 of the masked copy (or of dz);
    // ln_part[row block][2][N]: this tile's column sums of d xhat and d
    f32x4 re[4], xh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      re[q] = g.res ? *reinterpret_cast<const f32x4*>(g.res + (size_t)row * g.ldres + cb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      xh[q] = *reinterpret_cast<const f32x4*>(g.xhat + (size_t)row * N + cb + 8 * q);
    }
    const float rs = g.rstd[row];
    float dd[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dv = acc[4 * q + j] + re[q][j], gd = dv * ga[q][j];
        dd[4 * q + j] = dv; s1 += gd; s2 += gd * xh[q][j];
      }
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    g64_publish(mine, h ? s2 : s1, seq);
    G64_EMU_PUBLISHED();
    float pv[NG];
    const bool gotb = g64_collect<NG>(theirs, h, seq, pv, xerr, g.spin_max);
    G64_EMU_AGREE(gotb)
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < NG / 2; ++i) { t1 += pv[2 * i]; t2 += pv[2 * i + 1]; }
    const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
    const float m1 = (h ? o1 + t1 : t1 + o1) * invN, m2 = (h ? o2 + t2 : t2 + o2) * invN;       // (lower columns first on both lanes)
    const bool masked = g.C2 != nullptr || (g.C16 != nullptr && g.drop.thr != 0u && g.drop.st != nullptr);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v, vm;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = rs * (dd[4 * q + j] * ga[q][j] - m1 - xh[q][j] * m2);
        vm[j] = masked ? v[j] * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + cb + 8 * q + j)) : v[j];
      }
      *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + cb + 8 * q) = v;
      if (g.C2 != nullptr) *reinterpret_cast<f32x4*>(g.C2 + (size_t)row * g.ldc + cb + 8 * q) = vm;
      if (g.C16 != nullptr) {
        uint2 pk;
        pk.x = (uint32_t)gt_f2bf(vm[0]) | ((uint32_t)gt_f2bf(vm[1]) << 16); pk.y = (uint32_t)gt_f2bf(vm[2]) | ((uint32_t)gt_f2bf(vm[3]) << 16);
        *reinterpret_cast<uint2*>(g.C16 + (size_t)row * g.ldc16 + cb + 8 * q) = pk;
      }
    }
    // dgamma / dbeta partials of this 64 x 64 tile: (d xhat, d) through LDS [2][64 rows][64 + 1], 16-row runs summed per thread, then 4 runs
    __syncthreads();                                             // (every wave is past its last operand read)
    constexpr int TS = 65;
    float* const sg = smem, * const sb = smem + 64 * TS;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rr = wm * 32 + r32, cc = wn * 32 + 4 * h + 8 * q + j;
        sg[rr * TS + cc] = dd[4 * q + j] * xh[q][j];
        sb[rr * TS + cc] = dd[4 * q + j];
      }
    __syncthreads();
    const int tid = threadIdx.x, c = tid & 63, rg = tid >> 6;
    float ag = 0.f, ab = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ag += sg[(rg * 16 + r) * TS + c]; ab += sb[(rg * 16 + r) * TS + c]; }
    float* const sp = smem + 2 * 64 * TS;                        // [4 runs][2][64]
    sp[(rg * 2) * 64 + c] = ag; sp[(rg * 2 + 1) * 64 + c] = ab;
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6;
      const float v = (sp[(0 * 2 + which) * 64 + c] + sp[(1 * 2 + which) * 64 + c]) + (sp[(2 * 2 + which) * 64 + c] + sp[(3 * 2 + which) * 64 + c]);
      g.ln_part[((size_t)(m0 >> 6) * 2 + which) * N + n0 + c] = v;
    }
  }
}
// ================================================================================================================ the same on 128x128 tiles
// Round 6: the LayerNorm epilogues on the BIG tile (gemm32_kernel / gemm32h_kernel of gt_gemm32.h).  At d_model 512 / 16384 tokens (bs 512 on one
// GPU) an N = 512 Linear is 128 x 4 = 512 tiles of 128x128 -- exactly the two workgroups per CU the kernels keep resident, so the whole grid is
// co-resident there too -- and the norm was still a row pass of its own: 24 passes per step, 7.7 % of the fp32 step and 21 % of the bf16 one
// (3.7 GB of HBM traffic).  Same protocol as above, other geometry: a wave owns 64 rows x 64 columns (2 x 2 blocks of 32x32), so a PART is 64
// columns (N / 64 parts per row), a lane publishes one granule per row block ta (half 0: mean / sum g, half 1: M2 / sum g xhat of the wave's
// 64 columns) and combines N / 64 parts x 2 values per row, two rows per lane.  Granules of a 128-row block: [part][4 row quarters][2 values]
// [32 rows] -- a wave publishes 512 contiguous bytes per ta; the region is the one gt_rowx_floats sizes (this layout takes half of it).
// A shape takes ONE of the two geometries for good (by its tile count), so every granule in use still advances by exactly one per launch.
__device__ __forceinline__ unsigned long long* g128_rowx(const GemmArgs& g, const int m0) {
  return reinterpret_cast<unsigned long long*>(g.rowx + GT_ROWX_HDR) + (size_t)(m0 >> 7) * (g.N / 64) * 256;
}
// this wave's sequence numbers (one per ta), from the lane's OWN granules
__device__ __forceinline__ void g128_seq(const GemmArgs& g, const int m0, const int n0, const int wm, const int wn, const int r32, const int h, uint32_t (&seq)[2]) {
  const unsigned long long* q = g128_rowx(g, m0) + (size_t)((n0 >> 6) + wn) * 256 + (wm * 2) * 64 + h * 32 + r32;
  seq[0] = g64_seq(q); seq[1] = g64_seq(q + 64);
}
// both rows of a lane in ONE polling loop: parts [h NG / 2, (h + 1) NG / 2) of rows (quarter 2 wm + ta), (value 0, value 1) each
template <int NG>
__device__ __forceinline__ bool g128_collect(const unsigned long long* base, const int h, const uint32_t (&seq)[2], float (&v)[2][NG], unsigned* err, const int spin_max) {
  unsigned long long w[2][NG];
  bool ok = true;
  auto load_all = [&]() {
    ok = true;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int i = 0; i < NG / 2; ++i) {
        const unsigned long long* src = base + (size_t)(h * (NG / 2) + i) * 256 + ta * 64;      // part stride: [4 quarters][2 values][32 rows]
        w[ta][2 * i] = g64_ld(src); w[ta][2 * i + 1] = g64_ld(src + 32);
        ok = ok && (uint32_t)(w[ta][2 * i] >> 32) == seq[ta] && (uint32_t)(w[ta][2 * i + 1] >> 32) == seq[ta];
      }
  };
#ifdef GT_EMU
  load_all();
#else
  int spins = 0;
  for (;;) {
    load_all();
    if (__all(ok)) break;
    if (spins == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spins = spin_max;
    if (++spins > spin_max) { if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    __builtin_amdgcn_s_sleep(2);
  }
#endif
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int j = 0; j < NG; ++j) v[ta][j] = gt_u2f((uint32_t)w[ta][j]);
  return ok;
}
// NPH = N / 128 column tiles per row block (2: d_model 256, 4: d_model 512).  acc: the wave's 2 x 2 blocks in the store epilogue's lane map
// (lane (r32, h): row 32 ta + r32 of the wave's 64, registers 4 q + j of block tb = column 32 tb + 8 q + 4 h + j of its 64).
// smem: the operand buffers, free behind the main loop (>= 4 x 2080 + 256 floats: one [32][64 + 1] transpose area per wave + the meeting point).
template <int EPI, int NPH>
__device__ __forceinline__ void gemm32_ln_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], const int m0, const int n0, const int wm, const int wn,
                                                   const int r32, const int h, const uint32_t (&seq)[2], float* smem) {
  constexpr int N = 128 * NPH, NG = 2 * NPH;                  // NG: granules per lane and row = N / 64 parts, half of them per lane half, x 2 values
  const int row0 = m0 + wm * 64 + r32, cb = n0 + wn * 64 + 4 * h;          // rows row0 + 32 ta; columns cb + 32 tb + 8 q + j
  unsigned* const xerr = g.rowx;
  unsigned long long* const xrb = g128_rowx(g, m0);
  unsigned long long* const mine = xrb + (size_t)((n0 >> 6) + wn) * 256 + (wm * 2) * 64 + h * 32 + r32;       // + 64 ta
  const unsigned long long* const theirs = xrb + (wm * 2) * 64 + r32;                                          // + part * 256 + 64 ta (+ 32: value 1)
  const uint32_t dkey = gt_drop_key(g.drop);
  const float invN = 1.0f / (float)N;
  f32x4 ga[2][4];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int q = 0; q < 4; ++q) ga[tb][q] = *reinterpret_cast<const f32x4*>(g.gamma + cb + 32 * tb + 8 * q);
  if constexpr (EPI == EPI_RES_LN) {
    // z = drop(acc + bias) + res (left in acc);  y = LN(z) gamma + beta;  aux = xhat, aux2 = rstd, C16 = bf16(y)
    {
      f32x4 bi[2][4];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q) bi[tb][q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + cb + 32 * tb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        const int row = row0 + 32 * ta;
        f32x4 re[2][4];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            re[tb][q] = g.res ? *reinterpret_cast<const f32x4*>(g.res + (size_t)row * g.ldres + cb + 32 * tb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        float s = 0.f;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float lin = acc[ta][tb][4 * q + j] + bi[tb][q][j];
              if (g.round16) lin = gt_bf2f(gt_f2bf(lin));
              const float v = lin * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + cb + 32 * tb + 8 * q + j)) + re[tb][q][j];
              acc[ta][tb][4 * q + j] = v; s += v;
            }
        s += __shfl_xor(s, 32);
        const float mw = s * (1.0f / 64.0f);
        float qq = 0.f;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int e = 0; e < 16; ++e) { const float d = acc[ta][tb][e] - mw; qq += d * d; }
        qq += __shfl_xor(qq, 32);
        g64_publish(mine + 64 * ta, h ? qq : mw, seq[ta]);          // (lane half 0: the means, half 1: M2)
      }
    }
    G64_EMU_PUBLISHED();
    float pv[2][NG];
    const bool gotf = g128_collect<NG>(theirs, h, seq, pv, xerr, g.spin_max);
    G64_EMU_AGREE(gotf)
    f32x4 be[2][4];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int q = 0; q < 4; ++q) be[tb][q] = *reinterpret_cast<const f32x4*>(g.beta + cb + 32 * tb + 8 * q);
#pragma unroll
    for (int ta = 0; ta < 2; ++ta) {
      const int row = row0 + 32 * ta;
      // Chan's merge over this half's parts, in order (64 columns each), then the two halves, lower columns first on both lanes
      float mean = pv[ta][0], m2 = pv[ta][1], cnt = 64.f;
#pragma unroll
      for (int i = 1; i < NG / 2; ++i) {
        const float d = pv[ta][2 * i] - mean, n = cnt + 64.f;
        mean += d * (64.f / n);
        m2 += pv[ta][2 * i + 1] + d * d * (cnt * 64.f / n);
        cnt = n;
      }
      const float om = __shfl_xor(mean, 32), o2 = __shfl_xor(m2, 32);
      const float ma = h ? om : mean, a2 = h ? o2 : m2, mb = h ? mean : om, b2 = h ? m2 : o2;
      const float d = mb - ma;
      mean = ma + d * 0.5f;
      m2 = a2 + b2 + d * d * (cnt * 0.5f);
      const float rstd = 1.0f / sqrtf(m2 * invN + GT_LN_EPS);
      if (n0 == 0 && wn == 0 && h == 0) g.aux2[row] = rstd;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = cb + 32 * tb + 8 * q;
          f32x4 xh, yv;
#pragma unroll
          for (int j = 0; j < 4; ++j) { xh[j] = (acc[ta][tb][4 * q + j] - mean) * rstd; yv[j] = xh[j] * ga[tb][q][j] + be[tb][q][j]; }
          *reinterpret_cast<f32x4*>(g.aux + (size_t)row * N + col) = xh;
          *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = yv;
          if (g.C16 != nullptr) {
            uint2 pk;
            pk.x = (uint32_t)gt_f2bf(yv[0]) | ((uint32_t)gt_f2bf(yv[1]) << 16); pk.y = (uint32_t)gt_f2bf(yv[2]) | ((uint32_t)gt_f2bf(yv[3]) << 16);
            *reinterpret_cast<uint2*>(g.C16 + (size_t)row * g.ldc16 + col) = pk;
          }
        }
    }
  } else {
    // d = acc (+ res) (left in acc);  dz = LNbwd(d) with (xhat, rstd, gamma);  C2 = dz dropmask;  C16 = bf16 of the masked copy (or of dz);
    // ln_part[128-row block][2][N]: this tile's column sums of d xhat and d
    f32x4 xh[2][2][4];
    float rs[2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta) {
      const int row = row0 + 32 * ta;
      f32x4 re[2][4];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          re[tb][q] = g.res ? *reinterpret_cast<const f32x4*>(g.res + (size_t)row * g.ldres + cb + 32 * tb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
          xh[ta][tb][q] = *reinterpret_cast<const f32x4*>(g.xhat + (size_t)row * N + cb + 32 * tb + 8 * q);
        }
      rs[ta] = g.rstd[row];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float dv = acc[ta][tb][4 * q + j] + re[tb][q][j], gd = dv * ga[tb][q][j];
            acc[ta][tb][4 * q + j] = dv; s1 += gd; s2 += gd * xh[ta][tb][q][j];
          }
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      g64_publish(mine + 64 * ta, h ? s2 : s1, seq[ta]);
    }
    G64_EMU_PUBLISHED();
    float pv[2][NG];
    const bool gotb = g128_collect<NG>(theirs, h, seq, pv, xerr, g.spin_max);
    G64_EMU_AGREE(gotb)
    const bool masked = g.C2 != nullptr || (g.C16 != nullptr && g.drop.thr != 0u && g.drop.st != nullptr);
#pragma unroll
    for (int ta = 0; ta < 2; ++ta) {
      const int row = row0 + 32 * ta;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < NG / 2; ++i) { t1 += pv[ta][2 * i]; t2 += pv[ta][2 * i + 1]; }
      const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
      const float m1 = (h ? o1 + t1 : t1 + o1) * invN, m2 = (h ? o2 + t2 : t2 + o2) * invN;       // (lower columns first on both lanes)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = cb + 32 * tb + 8 * q;
          f32x4 v, vm;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = rs[ta] * (acc[ta][tb][4 * q + j] * ga[tb][q][j] - m1 - xh[ta][tb][q][j] * m2);
            vm[j] = masked ? v[j] * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + col + j)) : v[j];
          }
          *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = v;
          if (g.C2 != nullptr) *reinterpret_cast<f32x4*>(g.C2 + (size_t)row * g.ldc + col) = vm;
          if (g.C16 != nullptr) {
            uint2 pk;
            pk.x = (uint32_t)gt_f2bf(vm[0]) | ((uint32_t)gt_f2bf(vm[1]) << 16); pk.y = (uint32_t)gt_f2bf(vm[2]) | ((uint32_t)gt_f2bf(vm[3]) << 16);
            *reinterpret_cast<uint2*>(g.C16 + (size_t)row * g.ldc16 + col) = pk;
          }
        }
    }
    // dgamma / dbeta partials of this 128 x 128 tile: per wave, (d xhat, d) of its 64 x 64 block go through a [32 rows][64 + 1] area of its own
    // (4 passes: 2 quantities x 2 row blocks; lane c then sums column c over the 32 rows), the two row-waves of a column half meet in LDS
    __syncthreads();                                             // (every wave is past its last operand read)
    constexpr int TS = 65, WSZ = 32 * TS;
    float* const sw = smem + (wm * 2 + wn) * WSZ;
    const int lane = h * 32 + r32;
    float cg = 0.f, cbt = 0.f;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int ta = pass >> 1, which = pass & 1;               // which 0: d xhat (dgamma), 1: d (dbeta)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            sw[r32 * TS + 32 * tb + 8 * q + 4 * h + j] = which ? acc[ta][tb][4 * q + j] : acc[ta][tb][4 * q + j] * xh[ta][tb][q][j];
      GT_WAVE_SYNC();
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) a += sw[r * TS + lane];
      if (which) cbt += a; else cg += a;
      GT_WAVE_SYNC();
    }
    __syncthreads();
    float* const sp = smem + 4 * WSZ;                           // [2 column halves][2][64]: the lower row-wave's sums
    if (wm == 1) { sp[(wn * 2) * 64 + lane] = cg; sp[(wn * 2 + 1) * 64 + lane] = cbt; }
    __syncthreads();
    if (wm == 0) {
      g.ln_part[((size_t)(m0 >> 7) * 2) * N + n0 + wn * 64 + lane] = cg + sp[(wn * 2) * 64 + lane];
      g.ln_part[((size_t)(m0 >> 7) * 2 + 1) * N + n0 + wn * 64 + lane] = cbt + sp[(wn * 2 + 1) * 64 + lane];
    }
  }
}
// host side: N = d_model 256 / 512 and the whole grid of 128x128 tiles resident at once (two workgroups per CU), from the tile count at which the
// big tile is the Linear's kernel anyway (GT_T128_BIG_MIN; d_model 512: 8192 .. 16384 tokens, d_model 256: 12288 .. 32768)
#ifndef GT_LN128_MIN
#define GT_LN128_MIN GT_T128_BIG_MIN
#endif
static inline bool gemm32_ln_shape(const GemmArgs& g, int cus) {
  static const long lo = gt_env_long("GT_LN128_MIN", GT_LN128_MIN);
  const long tiles = (long)(g.M / 128) * (g.N / 128);
  return (g.N == 256 || g.N == 512) && g.M % 128 == 0 && tiles <= 2l * cus && tiles >= lo;
}
static inline bool gemm32_ln_ok(const GemmArgs& g, int epi) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!g.rowx || !g.gamma || !al16(g.gamma) || (g.res && ((g.ldres & 3) || !al16(g.res))) || !g.C || (g.ldc & 3) || !al16(g.C) || g.accumulate) return false;
  if (g.C16 && ((g.ldc16 & 3) || (reinterpret_cast<uintptr_t>(g.C16) & 7))) return false;
  if (epi == EPI_RES_LN && (!g.beta || !al16(g.beta) || !g.aux || !al16(g.aux) || !g.aux2 || (g.bias && !al16(g.bias)))) return false;
  if (epi == EPI_RES_LNBWD && (!g.xhat || !al16(g.xhat) || !g.rstd || !g.ln_part || (g.C2 && !al16(g.C2)))) return false;
  return true;
}

// ================================================================================================================ ... and on 32x32 tiles
// Round 6: the same row exchange under the GENERIC kernel's 32x32 tiles (gemm_kernel<2, 2, 1, 1, BK64>, 256 threads) -- the tile the Linears of
// the reference's d_model-256 YAMLs run on at 512 ... 2048 tokens, where every LayerNorm was a launch of its own at its 4.5 us floor (24 of
// the 91 launches of the K&S / Random YAML step, 44 of Random_test_large's).  Round 5 forced the 64x64 form there and lost (a quarter of the
// CUs got a tile); here the GEMM keeps its tiles and only the epilogue changes.  Geometry: a PART is the tile's 32 columns (N / 32 parts per
// row), a 16-lane group owns two of the tile's rows (two columns per lane); the tile's raw accumulators are staged in LDS first, the row
// partials meet in LDS, ONE wave publishes the tile's 64 granules ([part][2 values][32 rows]: 512 contiguous bytes), all four waves collect
// (thread t: part t >> 5, row t & 31) into LDS, every group merges the parts in order.  Granules of a 32-row block: N / 32 x 64 -- the
// region of gt_rowx_floats exactly.  A shape takes one geometry for good (ln_xchg_tile).
__device__ __forceinline__ unsigned long long* g32_rowx(const GemmArgs& g, const int m0) {
  return reinterpret_cast<unsigned long long*>(g.rowx + GT_ROWX_HDR) + (size_t)(m0 >> 5) * (g.N >> 5) * 64;
}
// this launch's sequence number: wave 0's lanes own the tile's granules (the others return 0 and take the number from LDS later)
__device__ __forceinline__ uint32_t gemm_xln32_tag(const GemmArgs& g, const int m0, const int n0) {
  const int tid = threadIdx.x;
  return tid < 64 ? g64_seq(g32_rowx(g, m0) + (size_t)(n0 >> 5) * 64 + tid) : 0u;
}
// smem: [32][36] raw accumulators of the tile (staged by the caller, barrier passed) + 3200 floats of scratch behind them
template <int EPI>
__device__ __forceinline__ void gemm_xln32_epilogue(const GemmArgs& g, const int m0, const int n0, const uint32_t tag0, float* smem) {
  constexpr int CSTR = 36;
  float* const sC = smem;
  float* const sS = smem + 32 * CSTR;            // [2 values][32 rows]: this tile's partials
  float* const sP = sS + 64;                     // [16 parts][2 values][32 rows]: every part's
  float* const sG = sP + 1024;                   // [16 groups][2][32 columns]: dgamma / dbeta partials of the groups
  uint32_t* const sTag = reinterpret_cast<uint32_t*>(sG + 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4, grp = wave * 4 + lg;
  const int N = g.N, NP = N >> 5, part = n0 >> 5;
  unsigned* const xerr = g.rowx;
  unsigned long long* const xrb = g32_rowx(g, m0);
  const uint32_t dkey = gt_drop_key(g.drop);
  const float invN = 1.0f / (float)N;
  if (tid == 0) *sTag = tag0;
  const int c0 = n0 + l16, c1 = n0 + l16 + 16;                       // this lane's two columns
  const float ga0 = g.gamma[c0], ga1 = g.gamma[c1];
  float z[2][2], xh[2][2], rs[2] = {0.f, 0.f};
  if constexpr (EPI == EPI_RES_LN_X) {
    const float bi0 = g.bias ? g.bias[c0] : 0.f, bi1 = g.bias ? g.bias[c1] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int rl = grp + 16 * rr, row = m0 + rl;
      const float r0 = g.res ? g.res[(size_t)row * g.ldres + c0] : 0.f, r1 = g.res ? g.res[(size_t)row * g.ldres + c1] : 0.f;
      float l0 = sC[rl * CSTR + l16] + bi0, l1 = sC[rl * CSTR + l16 + 16] + bi1;
      if (g.round16) { l0 = gt_bf2f(gt_f2bf(l0)); l1 = gt_bf2f(gt_f2bf(l1)); }
      z[rr][0] = l0 * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + c0)) + r0;
      z[rr][1] = l1 * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + c1)) + r1;
      const float mw = gt_red16(z[rr][0] + z[rr][1]) * (1.0f / 32.0f);
      const float d0 = z[rr][0] - mw, d1 = z[rr][1] - mw;
      const float qq = gt_red16(d0 * d0 + d1 * d1);
      if (l16 == 0) { sS[rl] = mw; sS[32 + rl] = qq; }
    }
  } else {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int rl = grp + 16 * rr, row = m0 + rl;
      const float r0 = g.res ? g.res[(size_t)row * g.ldres + c0] : 0.f, r1 = g.res ? g.res[(size_t)row * g.ldres + c1] : 0.f;
      xh[rr][0] = g.xhat[(size_t)row * N + c0]; xh[rr][1] = g.xhat[(size_t)row * N + c1];
      rs[rr] = g.rstd[row];
      z[rr][0] = sC[rl * CSTR + l16] + r0; z[rr][1] = sC[rl * CSTR + l16 + 16] + r1;
      const float g0 = z[rr][0] * ga0, g1 = z[rr][1] * ga1;
      const float s1 = gt_red16(g0 + g1), s2 = gt_red16(g0 * xh[rr][0] + g1 * xh[rr][1]);
      if (l16 == 0) { sS[rl] = s1; sS[32 + rl] = s2; }
    }
  }
  __syncthreads();
  const uint32_t seq = *sTag;
  if (wave == 0) g64_publish(xrb + (size_t)part * 64 + lane, sS[lane], seq);      // lanes 0..31: value 0 of rows 0..31, lanes 32..63: value 1
  G64_EMU_PUBLISHED();
  {
    // collect: thread t takes row t & 31 of parts t >> 5, (t >> 5) + 8, ... -- both values; the waves poll on their own
    const int r = tid & 31;
    unsigned long long w[2][2];
    bool ok = true;
    auto load_all = [&]() {
      ok = true;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pp = (tid >> 5) + 8 * k;
        if (pp < NP) {
          const unsigned long long* src = xrb + (size_t)pp * 64 + r;
          w[k][0] = g64_ld(src); w[k][1] = g64_ld(src + 32);
          ok = ok && (uint32_t)(w[k][0] >> 32) == seq && (uint32_t)(w[k][1] >> 32) == seq;
        }
      }
    };
#ifdef GT_EMU
    load_all();
#else
    int spins = 0;
    for (;;) {
      load_all();
      if (__all(ok)) break;
      if (spins == 0 && __hip_atomic_load(xerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spins = g.spin_max;
      if (++spins > g.spin_max) { if (lane == 0) __hip_atomic_store(xerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(2);
    }
#endif
    G64_EMU_AGREE(ok)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pp = (tid >> 5) + 8 * k;
      if (pp < NP) { sP[(pp * 2) * 32 + r] = gt_u2f((uint32_t)w[k][0]); sP[(pp * 2 + 1) * 32 + r] = gt_u2f((uint32_t)w[k][1]); }
    }
  }
  __syncthreads();
  if constexpr (EPI == EPI_RES_LN_X) {
    const float be0 = g.beta[c0], be1 = g.beta[c1];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int rl = grp + 16 * rr, row = m0 + rl;
      // Chan's merge over the parts, in order (32 columns each): identical statistics in every workgroup of the row
      float mean = sP[rl], m2 = sP[32 + rl], cnt = 32.f;
      for (int p = 1; p < NP; ++p) {
        const float d = sP[(p * 2) * 32 + rl] - mean, n = cnt + 32.f;
        mean += d * (32.f / n);
        m2 += sP[(p * 2 + 1) * 32 + rl] + d * d * (cnt * 32.f / n);
        cnt = n;
      }
      const float rstd = 1.0f / sqrtf(m2 * invN + GT_LN_EPS);
      if (part == 0 && l16 == 0) g.aux2[row] = rstd;
      const float x0 = (z[rr][0] - mean) * rstd, x1 = (z[rr][1] - mean) * rstd;
      const float y0 = x0 * ga0 + be0, y1 = x1 * ga1 + be1;
      g.aux[(size_t)row * N + c0] = x0; g.aux[(size_t)row * N + c1] = x1;
      g.C[(size_t)row * g.ldc + c0] = y0; g.C[(size_t)row * g.ldc + c1] = y1;
      if (g.C16 != nullptr) { g.C16[(size_t)row * g.ldc16 + c0] = gt_f2bf(y0); g.C16[(size_t)row * g.ldc16 + c1] = gt_f2bf(y1); }
    }
  } else {
    const bool masked = g.C2 != nullptr || (g.C16 != nullptr && g.drop.thr != 0u && g.drop.st != nullptr);
    float dg[2] = {0.f, 0.f}, db[2] = {0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int rl = grp + 16 * rr, row = m0 + rl;
      float t1 = 0.f, t2 = 0.f;
      for (int p = 0; p < NP; ++p) { t1 += sP[(p * 2) * 32 + rl]; t2 += sP[(p * 2 + 1) * 32 + rl]; }
      const float m1 = t1 * invN, m2 = t2 * invN;
      const float v0 = rs[rr] * (z[rr][0] * ga0 - m1 - xh[rr][0] * m2), v1 = rs[rr] * (z[rr][1] * ga1 - m1 - xh[rr][1] * m2);
      const float w0 = masked ? v0 * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + c0)) : v0;
      const float w1 = masked ? v1 * gt_drop_mul(g.drop, dkey, (uint32_t)(row * N + c1)) : v1;
      g.C[(size_t)row * g.ldc + c0] = v0; g.C[(size_t)row * g.ldc + c1] = v1;
      if (g.C2 != nullptr) { g.C2[(size_t)row * g.ldc + c0] = w0; g.C2[(size_t)row * g.ldc + c1] = w1; }
      if (g.C16 != nullptr) { g.C16[(size_t)row * g.ldc16 + c0] = gt_f2bf(w0); g.C16[(size_t)row * g.ldc16 + c1] = gt_f2bf(w1); }
      dg[0] += z[rr][0] * xh[rr][0]; dg[1] += z[rr][1] * xh[rr][1]; db[0] += z[rr][0]; db[1] += z[rr][1];
    }
    // dgamma / dbeta partials of this tile: the 16 groups' sums over their two rows meet in LDS, in a fixed order
    sG[(grp * 2) * 32 + l16] = dg[0]; sG[(grp * 2) * 32 + l16 + 16] = dg[1];
    sG[(grp * 2 + 1) * 32 + l16] = db[0]; sG[(grp * 2 + 1) * 32 + l16 + 16] = db[1];
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) a += sG[(q * 2 + which) * 32 + c];
      g.ln_part[((size_t)(m0 >> 5) * 2 + which) * N + n0 + c] = a;
    }
  }
}
// host side: the 32x32-tile form applies where a d_model-wide Linear runs on the generic kernel's 32x32 tiles anyway (fewer than GT_T64_MIN
// tiles of 64x64), N is a multiple of 32 up to 512, and the whole grid is resident with room to spare (<= 2 tiles per CU; at least half the CUs get one)
static inline bool gemm_xln32_shape(const GemmArgs& g, int cus) {
  static const long lo_env = gt_env_long("GT_LN32_MIN", -1);          // (tests lower the bound: the emulator runs small shapes)
  const long tiles = (long)(g.M / 32) * (g.N / 32), t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64), lo = lo_env >= 0 ? lo_env : cus / 2;
  return g.N % 32 == 0 && g.N >= 64 && g.N <= 512 && g.M % 32 == 0 && t64 < GT_T64_MIN && tiles <= 2l * cus && tiles >= lo;
}
static inline bool gemm_xln32_ok(const GemmArgs& g, int epi) {
  if (!g.rowx || !g.gamma || !g.C || g.accumulate) return false;
  if (epi == EPI_RES_LN_X && (!g.beta || !g.aux || !g.aux2)) return false;
  if (epi == EPI_RES_LNBWD_X && (!g.xhat || !g.rstd || !g.ln_part)) return false;
  return true;
}

// host side: the fused LayerNorm epilogues apply when the Linear itself can take the 64x64 kernels, N = d_model is 256 or 512 and the
// WHOLE grid is resident at once (two workgroups per CU)
// (forced: gt_set_ln_exchange(1) -- tests on small shapes: no lower bound on the tile count)
static inline bool gemm64_ln_shape(const GemmArgs& g, int cus, bool forced = false) {
  const long tiles = (long)(g.M / 64) * (g.N / 64);
  return (g.N == 256 || g.N == 512) && g.M % 64 == 0 && tiles <= 2l * cus && (forced || 4 * tiles >= 3l * cus);       // (and at least 3/4 of the CUs get a tile)
}

template <bool BKM, int EPI, int PREC = 0>
__global__ __launch_bounds__(256, 2) void gemm64_kernel(GemmArgs g) {
  typedef Gemm64Cfg Cfg;
  constexpr int BK = Cfg::BK, STR = Cfg::STR, SZ = Cfg::SZ, PER = 4;
  __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM];
  const int bid = gemm64_bid(), gx = gridDim.x;
  const int m0 = (bid / gx) * 64, n0 = (bid % gx) * 64, nk = g.K / BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  uint32_t xtag = 0;            // LayerNorm epilogues: this launch's sequence number = the wave's OWN ready word + 1 (read before the wave publishes)
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) xtag = g64_seq(g64_rowx(g, m0) + (size_t)((n0 >> 5) + wn) * 128 + wm * 64 + h * 32 + r32);
  uint32_t kbpre[1];
  gemm32_kbits_pre<EPI, 1, 1>(g, m0, n0, wm, wn, r32, h, kbpre);

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

  // (GT_G64_DEEP: the third register set of the three-deep ring -- slab t + 1 waits in registers, t + 2 and t + 3 are in flight)
  f32x4 ua[PER], ub[PER];
  auto kof = [&](const int t) { return (t < nk ? t : nk - 1) * BK; };
  G64_LD(va, vb, 0)
  G64_LD(wa, wb, BK)
  if constexpr (PREC == 0 && GT_G64_DEEP) { G64_LD(ua, ub, kof(2)) }
  G64_ST(va, vb, 0)
  if constexpr (PREC == 0 && GT_G64_DEEP) { G64_LD(va, vb, kof(3)) }
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
    // the same slab on the three-deep ring.  CUR: LDS buffer holding slab t; (NA, NB): registers holding slab t + 1 -- written to the other
    // buffer behind groups 0-2, then reloaded with slab t + 4 behind groups 3-5
#define G64_SLAB3(CUR, NA, NB, t)                                                              \
    if ((t) < nk) {                                                                            \
    G64_STA(NA, (CUR) ^ 1, 0) G64_STB(NB, (CUR) ^ 1, 0) G64_STA(NA, (CUR) ^ 1, 1)              \
    G64_RD(fa1, fb1, CUR, 1) G64_MM(fa0, fb0)                                                  \
    G64_STB(NB, (CUR) ^ 1, 1) G64_STA(NA, (CUR) ^ 1, 2) G64_STB(NB, (CUR) ^ 1, 2)              \
    G64_RD(fa0, fb0, CUR, 2) G64_MM(fa1, fb1)                                                  \
    G64_STA(NA, (CUR) ^ 1, 3) G64_STB(NB, (CUR) ^ 1, 3)                                        \
    G64_RD(fa1, fb1, CUR, 3) G64_MM(fa0, fb0)                                                  \
    G64_GRPW3() G64_GRPW3() G64_GRPW2() GT_SCHED_FENCE()                                       \
    { const int k4_ = kof((t) + 4);                                                            \
      G64_LD(NA, NB, k4_) }                                                                    \
    G64_RD(fa0, fb0, CUR, 4) G64_MM(fa1, fb1)                                                  \
    G64_RD(fa1, fb1, CUR, 5) G64_MM(fa0, fb0)                                                  \
    G64_RD(fa0, fb0, CUR, 6) G64_MM(fa1, fb1)                                                  \
    G64_RD(fa1, fb1, CUR, 7) G64_MM(fa0, fb0)                                                  \
    G64_GRP3(0x20) G64_GRP3(0x20) G64_GRP2(0x20) G64_GRP0() GT_SCHED_FENCE()                   \
    __syncthreads();                                                                           \
    G64_RD(fa0, fb0, (CUR) ^ 1, 0) G64_MM(fa1, fb1) G64_GRP0() GT_SCHED_FENCE()                \
    }
    if constexpr (GT_G64_DEEP) {
      for (int kt = 0; kt < nk; kt += 6) {
        G64_SLAB3(0, wa, wb, kt) G64_SLAB3(1, ua, ub, kt + 1) G64_SLAB3(0, va, vb, kt + 2)
        G64_SLAB3(1, wa, wb, kt + 3) G64_SLAB3(0, ua, ub, kt + 4) G64_SLAB3(1, va, vb, kt + 5)
      }
    } else {
      for (int kt = 0; kt < nk; kt += 2) {
        G64_SLAB(0, wa, wb, va, vb, kt)
        G64_SLAB(1, va, vb, wa, wb, kt + 1)
      }
    }
#undef G64_SLAB3
#undef G64_SLAB
#undef G64_RD
#undef G64_MM
  }
#undef G64_LD
#undef G64_ST
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) {
    if (g.N == 512) gemm64_ln_epilogue<EPI, 8>(g, acc[0][0], m0, n0, wm, wn, r32, h, xtag, smem);
    else            gemm64_ln_epilogue<EPI, 4>(g, acc[0][0], m0, n0, wm, wn, r32, h, xtag, smem);
  } else {
    gemm32_store_epilogue<EPI, 1, 1>(g, acc, m0, n0, wm, wn, r32, h, kbpre);
  }
}

// GT_TRACE_GEMM64=1: one line on stderr per launch (tests assert that a shape really took this kernel)
static inline void gemm64_trace(const char* what, const GemmArgs& g, bool bkm, int epi) {
  const char* e = getenv("GT_TRACE_GEMM64");                   // (read per launch: a test switches it on for some calls of its process; host side, ~0.1 us)
  const bool on = e && e[0] == '1';
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
  uint32_t xtag = 0;
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) xtag = g64_seq(g64_rowx(g, m0) + (size_t)((n0 >> 5) + wn) * 128 + wm * 64 + h * 32 + r32);
  uint32_t kbpre[1];
  gemm32_kbits_pre<EPI, 1, 1>(g, m0, n0, wm, wn, r32, h, kbpre);
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
  if constexpr (EPI == EPI_RES_LN || EPI == EPI_RES_LNBWD) {
    float* const fsm = reinterpret_cast<float*>(sm);             // (4 x 64 x 136 bf16 = 69632 bytes: room for the partials' 35 KB)
    if (g.N == 512) gemm64_ln_epilogue<EPI, 8>(g, acc[0][0], m0, n0, wm, wn, r32, h, xtag, fsm);
    else            gemm64_ln_epilogue<EPI, 4>(g, acc[0][0], m0, n0, wm, wn, r32, h, xtag, fsm);
  } else {
    gemm32_store_epilogue<EPI, 1, 1>(g, acc, m0, n0, wm, wn, r32, h, kbpre);
  }
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
