// Sequence-resident kernels for small models (SURVEY 7 "regime i": the reference's shipped YAMLs have d_model 32).
//
// One workgroup owns ONE sequence -- 32 tokens, the whole attention window -- and walks the entire encoder with it: every
// stage of every layer (in-proj, attention, out-proj, LayerNorm, FFN, LayerNorm ... output heads; and the whole chain back)
// runs in the same launch with workgroup barriers between stages.  No stage ever needs another sequence, so there is no
// grid-level synchronisation: the forward of the model is ONE launch and so is its backward (weight gradients still leave as
// the grouped dispatch of gt_gemm.h: they contract over all sequences).  At d_model 32 the step's 33-49 launches of ~5-8 us were
// nothing but launch / hand-over latency (0.6-1.3 % of the MFMA peak); the tile of a sequence is 4 KB and stays in LDS.
//
//   activations of the sequence   LDS: x, x1, ctx tiles [32][DP + 8], the qkv tile, the FFN tile [32][F + 8] (A operands, k contiguous)
//   weights                       read once per workgroup, straight from L2 into MFMA B fragments (no LDS staging: every wave
//                                 owns its output columns, so no weight element is needed twice) -- from FRAGMENT-ORDERED copies
//                                 that seq_pack_kernel writes in front of the forward: a wave's 16-byte-per-lane load is then
//                                 1 KB contiguous.  Loaded from the row-major matrix the same fragment is 16 rows x 64 bytes,
//                                 and one CU pulls only 15.7 B/clk that way against 57.8 B/clk for full lines
//                                 (tools/ubench/frag_load_bench.hip) -- at d_model 128 that, not the MFMA, set the stage time.
//   saved for backward / tests    written to the same workspace buffers the one-kernel-per-op path uses (gt_ws_find names)
//   attention                     transposed-score MFMA bodies (the scheme of gt_attn.h) on LDS operands, four heads at a time
//
// What bounds these kernels (in-kernel stamps, profiles/r02_final_seq_stamps_*.txt): a workgroup is a chain of ~8 dependent stages per
// layer and each stage is ONE wave's instruction stream per SIMD -- a wave64 VALU instruction occupies its SIMD for 4 cycles
// (8 with the SIMD's second wave), a quarter-rate v_mul_lo_u32 for 16, a dependent ds_bpermute / LDS access ~100.  So the rules
// here are instruction-count rules: 32-bit element offsets from wave-uniform bases (no 64-bit address arithmetic per load),
// 16-byte LDS / global accesses, wave-uniform branches instead of per-lane zero-page selects, every elementwise pass spread
// over all 512 threads (16 lanes per token row), row reductions by DPP (no LDS round trip), the dropout state fetched once.
//
// Three instantiations by d_model class DP = 32 / 64 / 128 (d_model % 16 == 0, <= DP).  At DP 128 (the headline workload:
// d_model 128, dim_feedforward 512) a sequence's layer is ~13 MFLOP: the matmul stages are bound by the fp32 MFMA rate of the
// ONE CU the workgroup runs on (256 FLOP/clk), the whole step is 7 launches instead of 49.  While twice the batch fits the CUs the
// SPLIT instantiations take over there: two workgroups per sequence, 16 token rows each, one launch per phase (see seq_fwd_kernel) --
// twice the CUs; round 3: the weight gradients ride in the backward phases' launches on the CUs that are still idle (gt_seq_wg.h) and
// the update writes the next step's weight packs: 9 launches per step.  In gt_train_step the launch that runs the output layer also
// computes the loss (fused tail).
//
// Supported: encoder-only, fp32 operands, d_model % 16 == 0 and <= 128, dim_feedforward % 16 == 0 and <= 512, src_dim <= 32,
// head_dim 16 / 32 / 64 or < 16 (seq_supported in groove_hip.hip).
#pragma once
#include "gt_seq_api.h"
#include <type_traits>

// In-kernel stamps (diagnostic build only; cdna_hip_programming.md 7): workgroup 0, thread 0 records the shader clock at stage
// boundaries into a buffer nothing else reads.  tools/seq_stamps.py prints the per-stage cycle counts.
#ifdef GT_SEQ_STAMPS
#define GT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(a.ws + a.stamps)[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
__shared__ long long* gt_sub_ptr;      // sub-stage stamps of ONE matmul stage (set around it by the stamped workgroup's thread 0)
#define GT_SUBSTAMP(i) do { if (threadIdx.x == 0 && gt_sub_ptr) gt_sub_ptr[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define GT_SUBWAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define GT_SUBSET(on) do { if (threadIdx.x == 0) gt_sub_ptr = ((on) && blockIdx.x == 0) ? reinterpret_cast<long long*>(a.ws + a.stamps) + 200 : nullptr; } while (0)
// every workgroup of forward phase 1: [shader clock, 100 MHz real-time clock] at its start and end -> int64 slots 1024 + 4 * blockIdx.x ..
#define GT_WGSTAMP(k) do { if (threadIdx.x == 0 && a.phase == 1 && blockIdx.x < 512) { long long* q_ = reinterpret_cast<long long*>(a.ws + a.stamps) + 1024 + 4 * blockIdx.x + 2 * (k); \
    q_[0] = (long long)__builtin_amdgcn_s_memtime(); q_[1] = (long long)__builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define GT_WGSTAMP(k) do { } while (0)
#define GT_STAMP(i) do { } while (0)
#define GT_SUBSTAMP(i) do { } while (0)
#define GT_SUBWAIT() do { } while (0)
#define GT_SUBSET(on) do { } while (0)
#endif

// Stage hand-overs inside these kernels go through LDS only (the global stores are copies for LATER kernels: saved activations,
// weight-gradient operands), so the stage barrier is GT_BARRIER(): it waits for this wave's LDS traffic, not for its
// outstanding global stores.
// Byte accounting (diagnostic builds only, tools/acct_writes.sh; WRONG RESULTS, measured with rocprofv3 --pmc WRITE_SIZE): GT_SEQ_ACCT bit 0 = the
// forward keeps only what the next launch reads (layer output, q / k / v) and skips the stores saved for the backward alone (P, ctx, x1, xhat1,
// hact, xhat2); bit 1 = the pair exchange's consumer does not re-zero what it read; bit 2 = no pair-exchange stores at all.
// Result (profiles/r06_c2_write_accounting.txt): of the 24.7 MB a forward launch of the headline step writes, 9.5 are saved activations, 4.2 the
// exchange's granules, 4.2 their re-zeroing, 6.8 the hand-over to the next launch (layer output, q / k / v) and the rest.
#ifndef GT_SEQ_ACCT
#define GT_SEQ_ACCT 0
#endif
#ifndef GT_SEQ_PFB_LN
#define GT_SEQ_PFB_LN 1    /* ... and the backward chain's LayerNorm operands at the phase's start (A/B switch) */
#endif
#ifndef GT_SEQ_PFLN128
#define GT_SEQ_PFLN128 0   /* d_model 128, SPLIT / QUAD kernels: the LayerNorm passes' small operands (bias / gamma / beta; backward: x-hat / rstd / gamma) requested ahead, as at
                              d_model 32 -- measured 0.3-0.5 % SLOWER on the headline (0.1991 vs 0.1982 ms, three interleaved bench pairs, profiles/r06_ab_pfln128.txt): off */
#endif
#ifndef GT_SEQ_PF64
#define GT_SEQ_PF64 1      /* ... and of d_model 64 (the reference CLI's default shape: 16 heads of 4) */
#endif
#ifndef GT_SEQ_PF32
#define GT_SEQ_PF32 1      /* d_model 32, SPLIT kernels: every stage's global operands requested a stage ahead (round 6); 0: as before */
#endif
#define GT_SEQ_WAVES 8
#define GT_SEQ_NT (GT_SEQ_WAVES * 64)
#define GT_SEQ_NT_WG GT_SEQ_NT
#include "gt_seq_wg.h"

// ---- dropout: the step state is read ONCE per kernel (three scalars); keys and multipliers are the ones of gt_common.h ------------
struct SeqDropK { uint32_t thr; float scale; uint32_t s_lo, s_hi; };
__device__ __forceinline__ SeqDropK seq_dropk(const SeqArgs& a) {
  SeqDropK k;
  k.thr = (a.st != nullptr) ? a.thr : 0u; k.scale = a.dscale; k.s_lo = 0u; k.s_hi = 0u;
  if (k.thr) { k.s_lo = a.st->seed_lo ^ gt_fmix32(a.st->step); k.s_hi = a.st->seed_hi; }
  return k;
}
__device__ __forceinline__ uint32_t seq_key(const SeqDropK& k, const int site) {       // == gt_drop_key
  if (!k.thr) return 0u;
  uint32_t x = k.s_lo ^ ((uint32_t)site * 0x9E3779B9u);
  x = gt_fmix32(x) ^ k.s_hi;
  return gt_fmix32(x + 0x7F4A7C15u);
}
__device__ __forceinline__ float seq_dmul(const SeqDropK& k, const uint32_t key, const uint32_t idx) {   // == gt_drop_mul
  if (!k.thr) return 1.0f;
  return ((gt_fmix32((idx * 0x9E3779B1u) ^ key) >> 8) >= k.thr) ? k.scale : 0.0f;
}
__device__ __forceinline__ bool seq_dkeep(const SeqDropK& k, const uint32_t key, const uint32_t idx) {   // k.thr != 0: is element idx kept
  return (gt_fmix32((idx * 0x9E3779B1u) ^ key) >> 8) >= k.thr;
}

// ---- sums over the 16 lanes of a token row (lanes 16 r .. 16 r + 15 of a wave = one DPP row): quad butterfly, then the mirrored
// half-row and row -- the same operand pairs as an xor-1/2/4/8 butterfly (bit-identical to it), without the LDS crossbar.
__device__ __forceinline__ float seq_row16_sum(float v) {
#ifdef GT_EMU
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
#else
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
#endif
  return v;
}

// ---- pair exchange between the two COLUMN PARTNERS of a QUAD forward (see seq_fwd_kernel): each of the two workgroups holds a partial
// [16][128] result (its half of a contraction) as one 16 x 16 accumulator tile per wave and needs the other's.  A value travels as an
// 8-byte granule {value bits, tag} written by ONE agent-scope store (sc1: write-through, never torn) and read by agent-scope loads that
// poll the granule itself -- no separate flag, no fence, no barrier: one fabric round trip (MI355X_MICROARCH.md, handoff-1to1 / Guideline
// 16 R2).  Slot layout per workgroup: [4 accumulator registers][512 threads] granules, so a wave-instruction moves 512 contiguous bytes;
// the partner's lane (wave, lane) reads exactly the granules its own (wave, lane) counterpart wrote.  The CONSUMER zeroes what it has
// read: every slot is written once and consumed once per launch, so the region is all zero between launches (gt_workspace_init zeroes
// it once) and a stale tag can never match.  The spin is bounded: a partner that never arrives (it would mean the two were not
// co-resident) raises the region's error word instead of hanging the GPU.
#define GT_XTAG 0x5EC0DE01u
#define GT_XCHG_WG_GRANULES (4 * GT_SEQ_NT)          /* 8-byte granules per workgroup slot (16 KB) */
// (GT_XCHG_SPIN_MAX, the default bound of the polling loop: gt_seq_api.h)
#ifdef GT_EMU
#define GT_XTAG_NOW (GT_XTAG + emu::launch_serial)     /* (the emulator re-runs a waiting workgroup: a per-launch tag instead of the reset) */
#else
#define GT_XTAG_NOW GT_XTAG
#endif
__device__ __forceinline__ void seq_xchg_put(unsigned long long* slot, const f32x4& v, const int tid) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned long long w = ((unsigned long long)GT_XTAG_NOW << 32) | (unsigned long long)gt_f2u(v[j]);
#ifdef GT_EMU
    slot[j * GT_SEQ_NT + tid] = w;
#else
    if (!(GT_SEQ_ACCT & 4)) __hip_atomic_store(slot + j * GT_SEQ_NT + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  }
}
__device__ __forceinline__ f32x4 seq_xchg_get(unsigned long long* slot, const int tid, unsigned* err, const int spin_max = GT_XCHG_SPIN_MAX) {
  unsigned long long w[4];
#ifdef GT_EMU
  bool ok = true;
  for (int j = 0; j < 4; ++j) { w[j] = slot[j * GT_SEQ_NT + tid]; ok = ok && (uint32_t)(w[j] >> 32) == GT_XTAG_NOW; }
  if (!ok) emu::block_retry();                               // the partner has not run yet: this workgroup is run again after it
#else
  // (a word raised by an EARLIER launch: one look, no spin -- gt_gemm64.h, g64_collect)
  int spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j] = __hip_atomic_load(slot + j * GT_SEQ_NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = ok && (uint32_t)(w[j] >> 32) == GT_XTAG;
    }
    if (__all(ok) || (GT_SEQ_ACCT & 4)) break;                 // (wave-uniform exit: the lanes of a wave leave together)
    if (spins == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spins = spin_max;       // (looked at only once the first poll failed)
    if (++spins > spin_max) { if ((tid & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (!(GT_SEQ_ACCT & 6)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) __hip_atomic_store(slot + j * GT_SEQ_NT + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  return f32x4{gt_u2f((uint32_t)w[0]), gt_u2f((uint32_t)w[1]), gt_u2f((uint32_t)w[2]), gt_u2f((uint32_t)w[3])};
}

// ================================================================================================================ matmul primitives
// C (32 x N) = A (32 x K, an LDS tile, k contiguous) * B, B from global memory straight into MFMA fragments.
//   BKM = false: B(k, n) = W[n * ldw + k] (Linear forward);  BKM = true: B(k, n) = W[k * ldw + n] (dgrad).
// A tile of 16 output columns: lane (l16, lg) holds rows l16 (acc0) and 16 + l16 (acc1), columns n0 + 4 lg + 0..3.
// Every global load of a stage is issued before the first MFMA (one memory round trip per stage).

// ---- edge version (the 27-wide input / output layers, once per kernel): any K <= 64, any N; masked through the zero page
template <bool BKM, bool VEC>
__device__ __forceinline__ void seq_ldb(float (&b)[4], const float* __restrict__ W, const int ldw, const int n, const int N, const int k,
                                        const int K, const float* zp) {
  if (!BKM && VEC) {               // K % 4 == 0, 16-byte rows: a float4 is in range or out as a whole
    const float4 v = *reinterpret_cast<const float4*>((n < N && k < K) ? W + (size_t)n * ldw + k : zp);
    b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
  } else if (!BKM) {               // rows of any length (the 27-wide symbolic input)
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *((n < N && k + j < K) ? W + (size_t)n * ldw + k + j : zp);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *((n < N && k + j < K) ? W + (size_t)(k + j) * ldw + n : zp);
  }
}
// HALF (the SPLIT kernels: a workgroup owns 16 of the 32 token rows): sA points at the first own row, only acc0 is computed
template <bool HALF>
__device__ __forceinline__ void seq_mma(f32x4& acc0, f32x4& acc1, const float (&b)[4], const float* sA, const int lda, const int l16, const int k) {
  const float4 a0 = *reinterpret_cast<const float4*>(sA + l16 * lda + k);
  acc0 = GT_MFMA16(b[0], a0.x, acc0); acc0 = GT_MFMA16(b[1], a0.y, acc0); acc0 = GT_MFMA16(b[2], a0.z, acc0); acc0 = GT_MFMA16(b[3], a0.w, acc0);
  if (!HALF) {
    const float4 a1 = *reinterpret_cast<const float4*>(sA + (16 + l16) * lda + k);
    acc1 = GT_MFMA16(b[0], a1.x, acc1); acc1 = GT_MFMA16(b[1], a1.y, acc1); acc1 = GT_MFMA16(b[2], a1.z, acc1); acc1 = GT_MFMA16(b[3], a1.w, acc1);
  }
}
// wave w owns tile w (N <= 128), K <= 16 NK.  epi(n0, acc0, acc1, bias4): bias4 = bias[n0 + 4 lg + 0..3] (zeros without a bias).
template <bool BKM, bool VEC, int NK, bool HALF, typename Epi>
__device__ __forceinline__ void seq_mm_edge(const float* sA, const int lda, const int K, const float* __restrict__ W, const int ldw, const int N,
                                            const float* __restrict__ bias, const int wave, const int lane, const float* zp, Epi epi) {
  const int l16 = lane & 15, lg = lane >> 4;
  const int n0 = wave * 16;
  if (n0 >= N) return;                                   // wave-uniform
  float b[NK][4], bi[4];
#pragma unroll
  for (int u = 0; u < NK; ++u) { if (16 * u < K) seq_ldb<BKM, VEC>(b[u], W, ldw, n0 + l16, N, 16 * u + 4 * lg, K, zp); }
#pragma unroll
  for (int r = 0; r < 4; ++r) bi[r] = *((bias != nullptr && n0 + 4 * lg + r < N) ? bias + n0 + 4 * lg + r : zp);
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NK; ++u) { if (16 * u < K) seq_mma<HALF>(acc0, acc1, b[u], sA, lda, l16, 16 * u + 4 * lg); }
  epi(n0, acc0, acc1, bi);
}

// ---- Fragment-ordered weights.  For a product C = A * B with B(k, n) taken from a weight matrix W (forward: B(k, n) = W[n][k];
// dgrad: B(k, n) = W[k][n]) the pack holds, for column tile t and k-step u (16 k each; nkt k-steps in the whole contraction),
// the 64 x 4 floats lane (l16, lg) feeds to four consecutive MFMAs:  pack[((t * nkt + u) * 64 + lane) * 4 + j] = B(16 u + 4 lg + j,
// 16 t + l16).  Both packs of a layer use one layout: [in_w | out_w | w1 | w2] at offsets 0, 3 d^2, 4 d^2, 4 d^2 + d F.
// FULL: nk == NK is known (d_model == DP, the common case): no per-k-step branches, so the compiler can count the outstanding loads
// (a next-tile prefetch stays in flight under the MFMAs instead of being drained by a vmcnt(0)).
template <int NK> struct SeqB { float4 v[NK]; };
template <int NK, bool FULL = false>
__device__ __forceinline__ void seq_b_load(SeqB<NK>& b, const float* __restrict__ Wp, const int nkt, const int t, const int u0, const int nk,
                                           const int lane) {
  const float* p = Wp + (unsigned)(((t * nkt + u0) * 64 + lane) * 4);
#pragma unroll
  for (int u = 0; u < NK; ++u) { if (FULL || u < nk) b.v[u] = *reinterpret_cast<const float4*>(p + 256 * u); }
}
#ifdef GT_SEQ_TU_FWD   // (non-template kernel: defined in the forward translation unit only)
// one wave per fragment: 4 fragments per 256-thread block, 2 packs x L layers x (3 d^2 + d^2 + 2 d F) / 256 fragments
__global__ __launch_bounds__(256) void seq_pack_kernel(SeqArgs a) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, lg = lane >> 4;
  const int d = a.d, F = a.F, d16 = d >> 4, f16 = F >> 4;
  const int nf0 = 3 * d16 * d16, nf1 = d16 * d16, nf2 = f16 * d16, T = nf0 + nf1 + 2 * nf2;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= 2 * a.L * T) return;
  const int pack = g / (a.L * T), r = g % (a.L * T), l = r / T;
  int f = r % T, R, C;
  int64_t src, moff;
  if (f < nf0) { src = a.p0.in_w; R = 3 * d; C = d; moff = 0; }
  else if (f < nf0 + nf1) { f -= nf0; src = a.p0.out_w; R = d; C = d; moff = (int64_t)3 * d * d; }
  else if (f < nf0 + nf1 + nf2) { f -= nf0 + nf1; src = a.p0.w1; R = F; C = d; moff = (int64_t)4 * d * d; }
  else { f -= nf0 + nf1 + nf2; src = a.p0.w2; R = d; C = F; moff = (int64_t)4 * d * d + (int64_t)d * F; }
  const float* W = a.prm + src + (int64_t)l * a.pstride;
  float4 v;
  if (pack == 0) {                                           // forward: B(k, n) = W[n][k], tiles over the rows of W
    const int nkt = C >> 4, t = f / nkt, u = f % nkt;
    v = *reinterpret_cast<const float4*>(W + (size_t)(16 * t + l16) * C + 16 * u + 4 * lg);
  } else {                                                   // dgrad: B(k, n) = W[k][n], tiles over the columns of W
    const int nkt = R >> 4, t = f / nkt, u = f % nkt;
    const float* p = W + (size_t)(16 * u + 4 * lg) * C + 16 * t + l16;
    v = make_float4(p[0], p[C], p[2 * (size_t)C], p[3 * (size_t)C]);
  }
  *reinterpret_cast<float4*>(a.ws + (pack ? a.pack_b : a.pack_f) + (int64_t)l * a.kstride + moff + (int64_t)f * 256 + lane * 4) = v;
}

// The optimizer update of the fused train step, folded with the packing of the NEXT step's weights (round 3: the update touches every
// weight anyway).  Part A, one wave per 16 x 16 block of a layer matrix (in_w, out_w, w1, w2 of every encoder layer): new weight ->
// the parameter buffer, the forward fragment (the block as loaded: lane (l16, lg) = row l16, columns 4 lg ..) and -- transposed through
// 1 KB of LDS -- the dgrad fragment; the gradient is consumed and zeroed.  Part B: every other parameter (biases, LayerNorms, input /
// output layer), the flat sweep of sgd_kernel / adam_kernel.  Same arithmetic as those kernels, element for element.
// err: the QUAD pair-exchange region's error word (nullptr: no such region).  A timed-out exchange (partner workgroups not co-resident: the
// launch went on with garbage partials) must never reach the parameters: with the word set the update leaves parameters and moments as
// they are -- only the consumed gradients are cleared and the weight copies rewritten -- and the word stays set until the host has seen it
// (StepEngine.check_exchange: zeroes the region, falls back to two workgroups per sequence).
struct SeqUpd { float* prm; float* g; float* m; float* v; int64_t n; const gt_step_state* st; int algo, step_advanced, nblk_a; const unsigned* err; };
__device__ __forceinline__ float seq_upd_elem(const SeqUpd& u, const int64_t i, const float w, const float g, const float k, const float b1,
                                              const float b2, const float step_size, const float inv_sqrt_bc2, const float gs, const float eps,
                                              const bool skip) {
  if (skip) return w;
  if (u.algo == 0) return w - k * g;
  const float gi = g * gs;
  const float mi = b1 * u.m[i] + (1.0f - b1) * gi;
  const float vi = b2 * u.v[i] + (1.0f - b2) * gi * gi;
  u.m[i] = mi; u.v[i] = vi;
  return w - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
}
__global__ __launch_bounds__(256) void seq_update_pack_kernel(SeqArgs a, SeqUpd u) {
  __shared__ float tr[4][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, lg = lane >> 4;
  const float k = u.st->lr * u.st->grad_scale;
  float b1 = 0.f, b2 = 0.f, step_size = 0.f, inv_sqrt_bc2 = 0.f, gs = u.st->grad_scale, eps = 0.f;
  if (u.algo == 1) {
    b1 = u.st->beta1; b2 = u.st->beta2; eps = u.st->eps;
    const float t = (float)(u.st->opt_step + (u.step_advanced ? 0u : 1u));
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    step_size = u.st->lr / bc1; inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  }
  // (the word: written through by an earlier launch of this stream; g[n - 1]: padding behind the 27-float output bias -- zero in a single
  //  process; a data-parallel host writes its error flag there before the gradient all-reduce, so that EVERY rank skips together)
  const bool skip = (u.err != nullptr && *u.err != 0u) || u.g[u.n - 1] != 0.f;
  const int d = a.d, F = a.F, d16 = d >> 4, f16 = F >> 4;
  const int nf0 = 3 * d16 * d16, nf1 = d16 * d16, nf2 = f16 * d16, T = nf0 + nf1 + 2 * nf2;
  if ((int)blockIdx.x < u.nblk_a) {
    const int gidx = blockIdx.x * 4 + wave;
    if (gidx >= a.L * T) return;
    const int l = gidx / T;
    int f = gidx % T, R, C;
    int64_t src, moff;
    if (f < nf0) { src = a.p0.in_w; R = 3 * d; C = d; moff = 0; }
    else if (f < nf0 + nf1) { f -= nf0; src = a.p0.out_w; R = d; C = d; moff = (int64_t)3 * d * d; }
    else if (f < nf0 + nf1 + nf2) { f -= nf0 + nf1; src = a.p0.w1; R = F; C = d; moff = (int64_t)4 * d * d; }
    else { f -= nf0 + nf1 + nf2; src = a.p0.w2; R = d; C = F; moff = (int64_t)4 * d * d + (int64_t)d * F; }
    const int nkt = C >> 4, br = f / nkt, bc = f % nkt;
    const int64_t e0 = src + (int64_t)l * a.pstride + (int64_t)(16 * br + l16) * C + 16 * bc + 4 * lg;
    const float4 w = *reinterpret_cast<const float4*>(u.prm + e0), g = *reinterpret_cast<const float4*>(u.g + e0);
    float4 nw;
    nw.x = seq_upd_elem(u, e0, w.x, g.x, k, b1, b2, step_size, inv_sqrt_bc2, gs, eps, skip);
    nw.y = seq_upd_elem(u, e0 + 1, w.y, g.y, k, b1, b2, step_size, inv_sqrt_bc2, gs, eps, skip);
    nw.z = seq_upd_elem(u, e0 + 2, w.z, g.z, k, b1, b2, step_size, inv_sqrt_bc2, gs, eps, skip);
    nw.w = seq_upd_elem(u, e0 + 3, w.w, g.w, k, b1, b2, step_size, inv_sqrt_bc2, gs, eps, skip);
    *reinterpret_cast<float4*>(u.prm + e0) = nw;
    *reinterpret_cast<float4*>(u.g + e0) = make_float4(0.f, 0.f, 0.f, 0.f);
    float* pf = a.ws + a.pack_f + (int64_t)l * a.kstride + moff;
    float* pb = a.ws + a.pack_b + (int64_t)l * a.kstride + moff;
    *reinterpret_cast<float4*>(pf + (int64_t)f * 256 + lane * 4) = nw;
    tr[wave][l16][4 * lg] = nw.x; tr[wave][l16][4 * lg + 1] = nw.y; tr[wave][l16][4 * lg + 2] = nw.z; tr[wave][l16][4 * lg + 3] = nw.w;
    GT_WAVE_SYNC();
    // dgrad fragment (tile over the columns of W = bc, k-step = br): lane (l16, lg) holds W[16 br + 4 lg + j][16 bc + l16]
    const float4 tv = make_float4(tr[wave][4 * lg][l16], tr[wave][4 * lg + 1][l16], tr[wave][4 * lg + 2][l16], tr[wave][4 * lg + 3][l16]);
    *reinterpret_cast<float4*>(pb + (int64_t)(bc * (R >> 4) + br) * 256 + lane * 4) = tv;
    return;
  }
  // part B: four consecutive floats per thread; tensors are 64-float aligned, so a quad lies inside a layer matrix or outside as a whole
  const int64_t i = ((int64_t)(blockIdx.x - u.nblk_a) * 256 + threadIdx.x) * 4;
  if (i >= u.n) return;
  const int64_t rel = i - a.p0.in_w;
  if (rel >= 0 && rel < (int64_t)a.L * a.pstride) {
    const int64_t r = rel % a.pstride;
    const int64_t o1 = a.p0.out_w - a.p0.in_w, o2 = a.p0.w1 - a.p0.in_w, o3 = a.p0.w2 - a.p0.in_w;
    if (r < (int64_t)3 * d * d || (r >= o1 && r < o1 + (int64_t)d * d) || (r >= o2 && r < o2 + (int64_t)d * F) || (r >= o3 && r < o3 + (int64_t)d * F)) return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (i + e < u.n - 1) {                                        // (n - 1: the guard element, neither updated nor cleared)
      u.prm[i + e] = seq_upd_elem(u, i + e, u.prm[i + e], u.g[i + e], k, b1, b2, step_size, inv_sqrt_bc2, gs, eps, skip);
      u.g[i + e] = 0.f;
    }
  }
}
#endif
// acc0 / acc1 (rows l16 / 16 + l16) += A[:, k0 .. k0 + 16 nk) * B; ap = sA + l16 * lda + k0 + 4 lg (the A fragments are re-read from
// acc0 / acc1 (rows l16 / 16 + l16 from ap's row) += A[:, k0 .. k0 + 16 nk) * B; the A fragments are re-read from LDS per tile
// (two 16-byte reads per 8 MFMAs).  HALF: only acc0.
template <int NK, bool FULL = false, bool HALF = false>
__device__ __forceinline__ void seq_b_mma(f32x4& acc0, f32x4& acc1, const SeqB<NK>& b, const float* ap, const int lda, const int nk) {
  if (FULL && HALF) {       // one accumulator: two interleaved partial sums would change the summation order, so one chain it is
    float4 a0 = *reinterpret_cast<const float4*>(ap);
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      float4 n0 = a0;
      if (u + 1 < NK) n0 = *reinterpret_cast<const float4*>(ap + 16 * (u + 1));
      acc0 = GT_MFMA16(b.v[u].x, a0.x, acc0); acc0 = GT_MFMA16(b.v[u].y, a0.y, acc0);
      acc0 = GT_MFMA16(b.v[u].z, a0.z, acc0); acc0 = GT_MFMA16(b.v[u].w, a0.w, acc0);
      GT_SCHED_FENCE()
      a0 = n0;
    }
    return;
  }
  if (HALF) {
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      if (u < nk) {
        const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * u);
        acc0 = GT_MFMA16(b.v[u].x, a0.x, acc0); acc0 = GT_MFMA16(b.v[u].y, a0.y, acc0);
        acc0 = GT_MFMA16(b.v[u].z, a0.z, acc0); acc0 = GT_MFMA16(b.v[u].w, a0.w, acc0);
      }
    }
    return;
  }
  if (FULL) {               // A fragments one k-step ahead of the MFMAs that use them; the fence keeps the compiler from hoisting all of them
    float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 16 * lda);
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      float4 n0 = a0, n1 = a1;
      if (u + 1 < NK) { n0 = *reinterpret_cast<const float4*>(ap + 16 * (u + 1)); n1 = *reinterpret_cast<const float4*>(ap + 16 * lda + 16 * (u + 1)); }
      acc0 = GT_MFMA16(b.v[u].x, a0.x, acc0); acc1 = GT_MFMA16(b.v[u].x, a1.x, acc1);
      acc0 = GT_MFMA16(b.v[u].y, a0.y, acc0); acc1 = GT_MFMA16(b.v[u].y, a1.y, acc1);
      acc0 = GT_MFMA16(b.v[u].z, a0.z, acc0); acc1 = GT_MFMA16(b.v[u].z, a1.z, acc1);
      acc0 = GT_MFMA16(b.v[u].w, a0.w, acc0); acc1 = GT_MFMA16(b.v[u].w, a1.w, acc1);
      GT_SCHED_FENCE()
      a0 = n0; a1 = n1;
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < NK; ++u) {
    if (FULL || u < nk) {
      const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * u), a1 = *reinterpret_cast<const float4*>(ap + 16 * lda + 16 * u);
      acc0 = GT_MFMA16(b.v[u].x, a0.x, acc0); acc1 = GT_MFMA16(b.v[u].x, a1.x, acc1);
      acc0 = GT_MFMA16(b.v[u].y, a0.y, acc0); acc1 = GT_MFMA16(b.v[u].y, a1.y, acc1);
      acc0 = GT_MFMA16(b.v[u].z, a0.z, acc0); acc1 = GT_MFMA16(b.v[u].z, a1.z, acc1);
      acc0 = GT_MFMA16(b.v[u].w, a0.w, acc0); acc1 = GT_MFMA16(b.v[u].w, a1.w, acc1);
    }
  }
}
// Short contraction (K % 16 == 0, <= 16 NK), N % 16 == 0: wave w owns tiles w, w + 8, ... (at most MAXT of them); the B fragment
// of the wave's next tile is requested before the MFMAs of the current one.  epi(n0, acc0, acc1, bias float4).
// ALL the fragments (and bias chunks) of a wave's tiles of a seq_mm_tiles stage in its all-at-once form, for a caller that requests them a
// stage ahead (d_model 32, round 6: seq_tiles_all_load before the stage in front, seq_mm_tiles_all in place of seq_mm_tiles)
template <int NK, int MAXT> struct SeqTilesAll { SeqB<NK> b[MAXT]; float4 bi[MAXT]; };
template <int NK, int MAXT, bool BIAS = true>
__device__ __forceinline__ void seq_tiles_all_load(SeqTilesAll<NK, MAXT>& f, const float* __restrict__ Wp, const int K, const int N,
                                                   const float* __restrict__ bias, const int wave, const int lane) {
  const int lg = lane >> 4, ntile = N >> 4, nk = K >> 4;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + i * GT_SEQ_WAVES, tc = t < ntile ? t : (wave < ntile ? wave : 0);     // (clamped: no branch around a load)
    seq_b_load<NK, true>(f.b[i], Wp, nk, tc, 0, nk, lane);
    if constexpr (BIAS) f.bi[i] = *reinterpret_cast<const float4*>(bias + 16 * tc + 4 * lg);     // (never a run-time test: a branch around this load cost a vmcnt(0) right behind it)
    else f.bi[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NK, int MAXT, bool HALF, typename Epi>
__device__ __forceinline__ void seq_mm_tiles_all(const float* sA, const int lda, const int K, const int N, const int wave, const int lane,
                                                 const SeqTilesAll<NK, MAXT>& f, Epi epi) {
  const int l16 = lane & 15, lg = lane >> 4, ntile = N >> 4, nk = K >> 4;
  if (wave >= ntile) return;                             // wave-uniform
  const float* ap = sA + l16 * lda + 4 * lg;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + i * GT_SEQ_WAVES;
    if (t < ntile) {
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      seq_b_mma<NK, true, HALF>(acc0, acc1, f.b[i], ap, lda, nk);
      epi(16 * t, acc0, acc1, f.bi[i]);
    }
  }
}
template <int NK, int MAXT, bool FULL, bool HALF, typename Epi>
__device__ __forceinline__ void seq_mm_tiles_impl(const float* sA, const int lda, const int K, const float* __restrict__ Wp, const int N,
                                                  const float* __restrict__ bias, const int wave, const int lane, Epi& epi,
                                                  const bool have_pre = false, const SeqB<NK> pre = SeqB<NK>(), const float4* pre_bias = nullptr) {
  const int l16 = lane & 15, lg = lane >> 4, ntile = N >> 4, nk = K >> 4;
  if (wave >= ntile) return;                             // wave-uniform
  const float* ap = sA + l16 * lda + 4 * lg;
  if constexpr (MAXT > 2 && NK * MAXT <= 8) {
    // short contraction, many tiles (d_model 32: FFN1 / FFN2 dgrad, 4 tiles of 2 k-steps per wave): ALL the wave's fragments are requested
    // before the first MFMA -- 8 x 16 bytes per lane -- so the stage pays one L2 round trip instead of one per tile (with a prefetch depth
    // of one tile the 8 MFMAs + epilogue of a tile cover a third of the next fragment's latency)
    SeqB<NK> ball[MAXT];
    float4 biall[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = wave + i * GT_SEQ_WAVES, tc = t < ntile ? t : wave;     // (clamped: no branch around a load)
      if (i == 0 && have_pre) ball[0] = pre;
      else seq_b_load<NK, FULL>(ball[i], Wp, nk, tc, 0, nk, lane);
      biall[i] = bias != nullptr ? *reinterpret_cast<const float4*>(bias + 16 * tc + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = wave + i * GT_SEQ_WAVES;
      if (t < ntile) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        seq_b_mma<NK, FULL, HALF>(acc0, acc1, ball[i], ap, lda, nk);
        epi(16 * t, acc0, acc1, biall[i]);
      }
    }
    return;
  }
  SeqB<NK> b[2];
  float4 bi[2];
  GT_SUBSTAMP(0);
  if (have_pre) b[0] = pre;                               // (workgroup-uniform: the wave's first fragment was requested a stage ahead)
  else seq_b_load<NK, FULL>(b[0], Wp, nk, wave, 0, nk, lane);
  if (have_pre && pre_bias != nullptr) bi[0] = *pre_bias;  // (the first tile's bias chunk came with its fragment)
  else bi[0] = bias != nullptr ? *reinterpret_cast<const float4*>(bias + 16 * wave + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
  GT_SUBSTAMP(1);
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + i * GT_SEQ_WAVES;
    if (t < ntile) {
      if (i + 1 < MAXT && t + GT_SEQ_WAVES < ntile) {
        seq_b_load<NK, FULL>(b[(i + 1) & 1], Wp, nk, t + GT_SEQ_WAVES, 0, nk, lane);
        bi[(i + 1) & 1] = bias != nullptr ? *reinterpret_cast<const float4*>(bias + 16 * (t + GT_SEQ_WAVES) + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      GT_SUBSTAMP(2 + 3 * i);
      seq_b_mma<NK, FULL, HALF>(acc0, acc1, b[i & 1], ap, lda, nk);
#ifdef GT_SEQ_STAMPS
      asm volatile("" : "+v"(acc0), "+v"(acc1));
#endif
      GT_SUBSTAMP(3 + 3 * i);
      epi(16 * t, acc0, acc1, bi[i & 1]);
      GT_SUBSTAMP(4 + 3 * i);
    }
  }
}
// FULL (K == 16 NK) is a property of the kernel instantiation (EXACT: d_model == DP): a run-time dispatch between the two bodies gets
// merged back into the branchy one by the compiler.
template <int NK, int MAXT, bool FULL, bool HALF, typename Epi>
__device__ __forceinline__ void seq_mm_tiles(const float* sA, const int lda, const int K, const float* __restrict__ Wp, const int N,
                                             const float* __restrict__ bias, const int wave, const int lane, Epi epi,
                                             const bool have_pre = false, const SeqB<NK> pre = SeqB<NK>(), const float4* pre_bias = nullptr) {
  seq_mm_tiles_impl<NK, MAXT, FULL, HALF>(sA, lda, K, Wp, N, bias, wave, lane, epi, have_pre, pre, pre_bias);
}
// the wave's first B fragment of a seq_mm_tiles stage (tile `wave`, all k-steps), for a caller that requests it a stage ahead
template <int NK>
__device__ __forceinline__ SeqB<NK> seq_tiles_first(const float* __restrict__ Wp, const int K, const int N, const int wave, const int lane) {
  SeqB<NK> b;
  seq_b_load<NK, true>(b, Wp, K >> 4, wave < (N >> 4) ? wave : 0, 0, K >> 4, lane);
  return b;
}
__device__ __forceinline__ void seq_mma4(f32x4& acc, const float4& b, const float4& a) {
  acc = GT_MFMA16(b.x, a.x, acc); acc = GT_MFMA16(b.y, a.y, acc); acc = GT_MFMA16(b.z, a.z, acc); acc = GT_MFMA16(b.w, a.w, acc);
}
// Square projection at d <= 64 (N = K = d): 2 * (d / 16) <= 8 units of (column tile, 16-row half), one per wave -- the raw 16 x 16
// results go to an LDS tile [32][srs]; bias / dropout / residual belong to the LayerNorm pass that reads it (all 512 threads).
__device__ __forceinline__ void seq_mm_square(const float* sA, const int lda, const int d, const float* __restrict__ Wp, float* sOut, const int srs,
                                              const int wave, const int lane) {
  const int l16 = lane & 15, lg = lane >> 4, t = wave >> 1, half = wave & 1;
  if (16 * t >= d) return;
  SeqB<4> b;
  float4 av[4];
  seq_b_load<4>(b, Wp, d >> 4, t, 0, d >> 4, lane);
#pragma unroll
  for (int u = 0; u < 4; ++u) { if (16 * u < d) av[u] = *reinterpret_cast<const float4*>(sA + (16 * half + l16) * lda + 16 * u + 4 * lg); }
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) { if (16 * u < d) seq_mma4(acc, b.v[u], av[u]); }
  *reinterpret_cast<float4*>(sOut + (16 * half + l16) * srs + 16 * t + 4 * lg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}
// Long contraction (K % 16 == 0, up to 512) into N <= 128 columns (NT = N / 16 tiles): the 8 waves split K as well -- wave w takes
// tile w % NT and k-part w / NT of KS = 8 / NT parts (NT 1, 2, 4, 8 -> KS 8, 4, 2, 1; other NT leave waves idle) -- and leave
// partial tiles in sR[part][32][srs]; the pass that reads them (seq_parts_sum, inside the following LayerNorm pass) sums the
// parts in a fixed order, part 0 first.  B moves in chunks of 8 k-steps, the next chunk requested before the current one's MFMAs.
__device__ __forceinline__ int seq_splitk_parts(const int N) { return GT_SEQ_WAVES / (N >> 4); }
// The wave's FIRST B chunk of a seq_mm_splitk stage, requested by the caller before the barrier that opens the stage (the weights do not
// depend on the stage before: when that one is a matmul too -- FFN1 -> FFN2, FFN2 dgrad -> FFN1 dgrad -- nothing but its epilogue and the
// barrier lies between, and the 1.8 k cycles to the first fragment disappear behind them).  Whole-chunk shapes only (seq_splitk_pre_ok).
__device__ __forceinline__ bool seq_splitk_pre_ok(const int K, const int N) {
  const int NT = N >> 4, KS = NT > 0 && NT <= GT_SEQ_WAVES ? GT_SEQ_WAVES / NT : 0, nks = K >> 4;
  return KS > 0 && nks % (8 * KS) == 0;                 // every part a whole number of 8-k-step chunks
}
__device__ __forceinline__ SeqB<8> seq_splitk_first(const float* __restrict__ Wp, const int K, const int N, const int wave, const int lane) {
  const int NT = N >> 4, KS = GT_SEQ_WAVES / NT, t = wave % NT, part = wave / NT, nks = K >> 4, per = nks / KS;
  SeqB<8> b;
  seq_b_load<8, true>(b, Wp, nks, t, (part < KS ? part : 0) * per, 8, lane);
  return b;
}
// acc0 / acc1 += A[:, 16 ks0 .. 16 ks1) * B for column tile t of a product whose WHOLE contraction has nkt k-steps (the pack's stride):
// B in chunks of 8 k-steps, the next chunk requested before the current one's MFMAs; pre: the first chunk, requested by the caller
template <bool HALF>
__device__ __forceinline__ void seq_mm_krange(f32x4& acc0, f32x4& acc1, const float* ap, const int lda, const float* __restrict__ Wp, const int nkt,
                                              const int t, const int ks0, const int ks1, const int lane, const bool have_pre, const SeqB<8> pre) {
  SeqB<8> b[2];
  if (ks1 > ks0 && ((ks1 - ks0) & 7) == 0) {                       // whole chunks only (F and 3 d multiples of 128 per part): branch-free bodies
    if (have_pre) b[0] = pre;                                     // (workgroup-uniform)
    else seq_b_load<8, true>(b[0], Wp, nkt, t, ks0, 8, lane);
    for (int c0 = ks0; c0 < ks1; c0 += 16) {                       // two chunks per trip: the buffer index stays compile-time
      if (c0 + 8 < ks1) seq_b_load<8, true>(b[1], Wp, nkt, t, c0 + 8, 8, lane);
      seq_b_mma<8, true, HALF>(acc0, acc1, b[0], ap + 16 * c0, lda, 8);
      if (c0 + 8 < ks1) {
        if (c0 + 16 < ks1) seq_b_load<8, true>(b[0], Wp, nkt, t, c0 + 16, 8, lane);
        seq_b_mma<8, true, HALF>(acc0, acc1, b[1], ap + 16 * (c0 + 8), lda, 8);
      }
    }
  } else {
    if (ks0 < ks1) seq_b_load<8>(b[0], Wp, nkt, t, ks0, ks1 - ks0, lane);
    for (int c0 = ks0; c0 < ks1; c0 += 16) {
      if (c0 + 8 < ks1) seq_b_load<8>(b[1], Wp, nkt, t, c0 + 8, ks1 - c0 - 8, lane);
      seq_b_mma<8, false, HALF>(acc0, acc1, b[0], ap + 16 * c0, lda, ks1 - c0);
      if (c0 + 8 < ks1) {
        if (c0 + 16 < ks1) seq_b_load<8>(b[0], Wp, nkt, t, c0 + 16, ks1 - c0 - 16, lane);
        seq_b_mma<8, false, HALF>(acc0, acc1, b[1], ap + 16 * (c0 + 8), lda, ks1 - c0 - 8);
      }
    }
  }
}
template <bool HALF>
__device__ __forceinline__ void seq_mm_splitk(const float* sA, const int lda, const int K, const float* __restrict__ Wp, const int N,
                                              float* sR, const int srs, const int wave, const int lane, const bool have_pre = false,
                                              const SeqB<8> pre = SeqB<8>()) {
  const int l16 = lane & 15, lg = lane >> 4;
  const int NT = N >> 4, KS = GT_SEQ_WAVES / NT;
  const int t = wave % NT, part = wave / NT;
  if (part >= KS) return;
  const int nks = K >> 4, per = (nks + KS - 1) / KS, ks0 = part * per, ks1 = (ks0 + per < nks) ? ks0 + per : nks;
  const int n0 = t * 16;
  const float* ap = sA + l16 * lda + 4 * lg;
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
  seq_mm_krange<HALF>(acc0, acc1, ap, lda, Wp, nks, t, ks0, ks1, lane, have_pre, pre);
  float* r = sR + part * 32 * srs + n0 + 4 * lg;
  *reinterpret_cast<float4*>(r + l16 * srs) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
  if (!HALF) *reinterpret_cast<float4*>(r + (16 + l16) * srs) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
}

// ================================================================================================================ row passes
// Elementwise / LayerNorm passes over a [32][d] tile use ALL 512 threads: thread (row = tid >> 4, seg = tid & 15) owns the
// CW = DP / 16 columns seg * CW ..; the 16 lanes of a row are one DPP row.  d % 16 == 0, so a thread's columns are in range
// (c0 < d) or out as a whole.
template <int CW> struct SeqVec;
template <> struct SeqVec<2> {
  static __device__ __forceinline__ void ld(float (&v)[2], const float* p) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <> struct SeqVec<4> {
  static __device__ __forceinline__ void ld(float (&v)[4], const float* p) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct SeqVec<8> {
  static __device__ __forceinline__ void ld(float (&v)[8], const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p), u = *reinterpret_cast<const float4*>(p + 4);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
// sum of the split-K partial tiles of this thread's columns, part 0 first
template <int CW>
__device__ __forceinline__ void seq_parts_sum(float (&v)[CW], const float* sR, const int srs, const int parts, const int row, const int c0) {
  SeqVec<CW>::ld(v, sR + row * srs + c0);
  for (int p = 1; p < parts; ++p) {
    float u[CW];
    SeqVec<CW>::ld(u, sR + (p * 32 + row) * srs + c0);
#pragma unroll
    for (int e = 0; e < CW; ++e) v[e] += u[e];
  }
}

// a [32][ncol] LDS tile -> the sequence's rows in global memory, 16 bytes per thread and pass: every wave-instruction writes 1 KB of
// consecutive addresses.  (The MFMA epilogues do NOT store their tiles to global themselves: a fragment-shaped store is 16 rows x
// 64 bytes per instruction and moves ~15 B/clk per CU, like the fragment-shaped loads -- tools/ubench/frag_load_bench.hip.)
// rows rb .. rb + nrows - 1 of the tile (dst: the sequence's row 0)
__device__ __forceinline__ void seq_tile_out(float* __restrict__ dst, const float* sT, const int str, const int ncol, const int tid, const int rb = 0,
                                             const int nrows = 32) {
  const int q4 = ncol >> 2;
  for (int e = tid; e < nrows * q4; e += GT_SEQ_NT) {
    const int r = rb + e / q4, c = (e % q4) * 4;
    *reinterpret_cast<float4*>(dst + (unsigned)(r * ncol + c)) = *reinterpret_cast<const float4*>(sT + r * str + c);
  }
}

// columns [c0, c0 + ncol) of rows rb .. rb + nrows - 1 of the tile -> the same columns of the sequence's rows (row stride ldd floats)
__device__ __forceinline__ void seq_tile_out_cols(float* __restrict__ dst, const int ldd, const float* sT, const int str, const int c0, const int ncol,
                                                  const int tid, const int rb, const int nrows) {
  const int q4 = ncol >> 2;
  for (int e = tid; e < nrows * q4; e += GT_SEQ_NT) {
    const int r = rb + e / q4, c = c0 + (e % q4) * 4;
    *reinterpret_cast<float4*>(dst + (unsigned)(r * ldd + c)) = *reinterpret_cast<const float4*>(sT + r * str + c);
  }
}

// LayerNorm forward: z (this thread's CW values, from zfun(row, c0, z)) -> y = LN(z) gamma + beta -> the LDS tile sY and the
// global y / xhat / rstd rows of this sequence (gy, gxhat, grstd: wave-uniform bases of the sequence's first row)
// gamma / beta / the Linear's bias of a thread's columns, requested a stage ahead of the LayerNorm pass that uses them (d_model 32, round 6)
template <int CW> struct SeqLnPre { float ga[CW], be[CW], bi[CW]; };
template <int CW>
__device__ __forceinline__ void seq_ln_pre(SeqLnPre<CW>& p, const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                           const int tid) {
  const int c0 = (tid & 15) * CW;
  SeqVec<CW>::ld(p.ga, gamma + c0); SeqVec<CW>::ld(p.be, beta + c0); SeqVec<CW>::ld(p.bi, bias + c0);
}
template <int DP, bool HALF, typename ZFun>
__device__ __forceinline__ void seq_ln_fwd(ZFun zfun, float* sY, const int str, const int d, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float* gy, float* gxhat, float* grstd, const int tid, const int rb,
                                           const SeqLnPre<DP / 16>* pre = nullptr) {
  constexpr int CW = DP / 16;
  if (HALF && tid >= 256) return;                            // 16 own rows: waves 0..3
  const int row = rb + (tid >> 4), seg = tid & 15, c0 = seg * CW;
  const bool ok = c0 < d;
  float z[CW], ga[CW], be[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) { z[e] = 0.f; ga[e] = 0.f; be[e] = 0.f; }
  if (ok) {
    if (pre != nullptr) {
#pragma unroll
      for (int e = 0; e < CW; ++e) { ga[e] = pre->ga[e]; be[e] = pre->be[e]; }
    } else { SeqVec<CW>::ld(ga, gamma + c0); SeqVec<CW>::ld(be, beta + c0); }
    zfun(row, c0, z);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < CW; ++e) s += z[e];
  const float invd = 1.0f / (float)d, mean = seq_row16_sum(s) * invd;
  float q = 0.f;
  if (ok) {
#pragma unroll
    for (int e = 0; e < CW; ++e) { const float t = z[e] - mean; q += t * t; }
  }
  const float rs = 1.0f / sqrtf(seq_row16_sum(q) * invd + GT_LN_EPS);
  if (ok) {
    float xh[CW], y[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) { xh[e] = (z[e] - mean) * rs; y[e] = xh[e] * ga[e] + be[e]; }
    const unsigned o = (unsigned)(row * d + c0);
    SeqVec<CW>::st(sY + row * str + c0, y);
    if (gy != nullptr) {                                      // (wave-uniform; nullptr: another workgroup saves these rows -- QUAD)
      SeqVec<CW>::st(gy + o, y);
      if (!(GT_SEQ_ACCT & 1)) SeqVec<CW>::st(gxhat + o, xh);
    }
  }
  if (seg == 0 && gy != nullptr) grstd[row] = rs;
}

// the LayerNorm backward's saved operands of a thread -- x-hat of its columns, rstd of its row, gamma -- requested ahead (d_model 32, round 6:
// at the start of the backward phase, two stages and an attention backward before their pass); 16 own rows: threads 0..255 (the others' copy is unused)
template <int CW> struct SeqLnBwdPre { float xh[CW], ga[CW], rs; };
template <int CW>
__device__ __forceinline__ void seq_ln_bwd_pre(SeqLnBwdPre<CW>& p, const int d, const float* __restrict__ gxhat, const float* __restrict__ grstd,
                                               const float* __restrict__ gamma, const int tid, const int rb) {
  const int row = rb + ((tid >> 4) & 15), c0 = (tid & 15) * CW;
  SeqVec<CW>::ld(p.xh, gxhat + (unsigned)(row * d + c0)); SeqVec<CW>::ld(p.ga, gamma + c0);
  p.rs = grstd[row];
}
// LayerNorm backward: g (from gfun) -> dz = LNbwd(g) -> sDz (LDS, unmasked: the residual gradient), dz * dropout mask -> sDzm
// (LDS: the next dgrad's A operand), both to global when gdz / gdzm are given (weight-gradient operands); the per-wave column
// sums of g xhat / g (4 rows each) -> sP[wave][2][DP]; seq_ln_part sums them over the waves after the stage barrier.
template <int DP, bool HALF, typename GFun>
__device__ __forceinline__ void seq_ln_bwd(GFun gfun, float* sDz, float* sDzm, const int str, const int d, const float* __restrict__ gxhat,
                                           const float* __restrict__ grstd, const float* __restrict__ gamma, const SeqDropK& dk, const uint32_t key,
                                           const uint32_t idx0, float* gdz, float* gdzm, float* sP, const int tid, const int rb,
                                           const SeqLnBwdPre<DP / 16>* pre = nullptr) {
  constexpr int CW = DP / 16;
  if (HALF && tid >= 256) return;                            // 16 own rows: waves 0..3
  const int row = rb + (tid >> 4), seg = tid & 15, c0 = seg * CW, lane = tid & 63, wave = tid >> 6;
  const bool ok = c0 < d;
  const unsigned o = (unsigned)(row * d + c0);
  float g[CW], xh[CW], ga[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) { g[e] = 0.f; xh[e] = 0.f; ga[e] = 0.f; }
  float rs;
  if (pre != nullptr) {                                      // (workgroup-uniform) saved x-hat / rstd / gamma requested at the phase's start
    rs = pre->rs;
#pragma unroll
    for (int e = 0; e < CW; ++e) { xh[e] = pre->xh[e]; ga[e] = pre->ga[e]; }
    if (ok) gfun(row, c0, g);
  } else {
    rs = grstd[row];
    if (ok) { SeqVec<CW>::ld(xh, gxhat + o); SeqVec<CW>::ld(ga, gamma + c0); gfun(row, c0, g); }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < CW; ++e) { const float gd = g[e] * ga[e]; s1 += gd; s2 += gd * xh[e]; }
  const float invd = 1.0f / (float)d, m1 = seq_row16_sum(s1) * invd, m2 = seq_row16_sum(s2) * invd;
  float cg[CW], cb[CW];                                          // column sums over the 4 rows of this wave
#pragma unroll
  for (int e = 0; e < CW; ++e) { cg[e] = g[e] * xh[e]; cb[e] = g[e]; }
#pragma unroll
  for (int e = 0; e < CW; ++e) { cg[e] += __shfl_xor(cg[e], 16); cb[e] += __shfl_xor(cb[e], 16); }
#pragma unroll
  for (int e = 0; e < CW; ++e) { cg[e] += __shfl_xor(cg[e], 32); cb[e] += __shfl_xor(cb[e], 32); }
  if (lane < 16 && ok) { SeqVec<CW>::st(sP + (wave * 2) * DP + c0, cg); SeqVec<CW>::st(sP + (wave * 2 + 1) * DP + c0, cb); }
  if (ok) {
    float v[CW], vm[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) {
      v[e] = rs * (g[e] * ga[e] - m1 - xh[e] * m2);
      vm[e] = v[e] * seq_dmul(dk, key, idx0 + o + e);
    }
    SeqVec<CW>::st(sDz + row * str + c0, v);
    SeqVec<CW>::st(sDzm + row * str + c0, vm);
    if (gdz) SeqVec<CW>::st(gdz + o, v);
    if (gdzm) SeqVec<CW>::st(gdzm + o, vm);
  }
}
// ... after the barrier: dgamma / dbeta partials of this sequence -> part[2][d], waves summed in a fixed order.  Runs on the LAST
// 2 * DP threads of the workgroup (the waves with the least matmul work in the stage that follows).
template <int DP, bool HALF>
__device__ __forceinline__ void seq_ln_part(const float* sP, float* part, const int d, const int tid) {
  const int t = tid - (GT_SEQ_NT - 2 * DP);
  if (t < 0) return;
  const int which = t / DP, c = t % DP;
  if (c >= d) return;
  float s = sP[which * DP + c];
#pragma unroll
  for (int w = 1; w < (HALF ? GT_SEQ_WAVES / 2 : GT_SEQ_WAVES); ++w) s += sP[(2 * w + which) * DP + c];
  part[which * d + c] = s;
}

// ================================================================================================================ attention
// The transposed-score scheme of gt_attn.h (S^T = K Q^T so that softmax rows are in-lane and P feeds the next MFMA without any
// data movement) on LDS operands: q / k / v (and dctx in the backward) are tiles of this workgroup, read with 32-bit LDS
// addresses.  Wave pair p = wave >> 1 takes head h4 + p, wave & 1 the query (role 2: key) tile.
// HDC: head-dim class the kernel is compiled for -- 0: head_dim < 16 (operands zero-padded to 16 columns), else 16 / 32 / 64
template <int HDC> struct SeqHd { static constexpr int HD = HDC ? HDC : 16; static constexpr bool PAD = HDC == 0; };
// PAD: columns >= head_dim read as 0 (the address is clamped to the head's first column: always inside the tile)
template <bool PAD>
__device__ __forceinline__ float4 seq_ld4(const float* p, const int col, const int hd) {
  if (!PAD) return *reinterpret_cast<const float4*>(p);
  float4 v;
  v.x = col + 0 < hd ? p[0] : 0.f; v.y = col + 1 < hd ? p[col + 1 < hd ? 1 : 0] : 0.f;
  v.z = col + 2 < hd ? p[col + 2 < hd ? 2 : 0] : 0.f; v.w = col + 3 < hd ? p[col + 3 < hd ? 3 : 0] : 0.f;
  if (!(col < hd)) v.x = 0.f;
  return v;
}
template <bool PAD>
__device__ __forceinline__ float seq_ld1(const float* p, const int col, const int hd) {
  if (!PAD) return *p;
  const float v = p[col < hd ? 0 : -col];
  return col < hd ? v : 0.f;
}
struct SeqAttn {
  const float* q; const float* k; const float* v; int ldq;      // LDS tiles (row stride ldq), already offset to the head's first column
  float* P; uint32_t pidx;                                       // this head's probabilities (global, [32][32]) and its dropout index base
  int hd; float scale;
};
template <int HD, bool PAD>
__device__ __forceinline__ void seq_attn_fwd(const SeqAttn& a, float* ctx, const int ldc, const SeqDropK& dk, const uint32_t key, const int ti,
                                             const int lane) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  const int i = 16 * ti + l16;                                   // this lane's query row
  const float* qrow = a.q + i * a.ldq + (PAD ? 0 : 4 * g);
  const float* krow = a.k + l16 * a.ldq + (PAD ? 0 : 4 * g);     // key tile 0; tile 1 = + 16 rows
  float4 qf[NQ], k0[NQ], k1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (!PAD) {
      qf[q] = *reinterpret_cast<const float4*>(qrow + 16 * q);
      k0[q] = *reinterpret_cast<const float4*>(krow + 16 * q);
      k1[q] = *reinterpret_cast<const float4*>(krow + 16 * a.ldq + 16 * q);
    } else {                                                      // head_dim < 16: element 4 g + j of the head, 0 beyond it
      float t[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j, cc = c < hdr ? c : 0;
        t[0][j] = c < hdr ? qrow[cc] : 0.f; t[1][j] = c < hdr ? krow[cc] : 0.f; t[2][j] = c < hdr ? krow[16 * a.ldq + cc] : 0.f;
      }
      qf[q] = make_float4(t[0][0], t[0][1], t[0][2], t[0][3]);
      k0[q] = make_float4(t[1][0], t[1][1], t[1][2], t[1][3]);
      k1[q] = make_float4(t[2][0], t[2][1], t[2][2], t[2][3]);
    }
  }
  float vb[NQ][2][4];                                             // V[4g + c + 16 tj][16 ct + l16]
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 16 * ct + l16, cc = (!PAD || col < hdr) ? col : 0;
        const float v = a.v[(4 * g + c + 16 * tj) * a.ldq + cc];
        vb[ct][tj][c] = (!PAD || col < hdr) ? v : 0.f;
      }
  f32x4 st[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};           // S^T tiles [tj]
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
    for (int r = 0; r < 4; ++r) { sv[tj][r] = st[tj][r] * a.scale; mx = fmaxf(mx, sv[tj][r]); }
  mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) { sv[tj][r] = expf(sv[tj][r] - mx); sum += sv[tj][r]; }
  sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  float pd[2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const unsigned o = (unsigned)(i * 32 + 16 * tj + 4 * g);
    float4 pv;
    pv.x = sv[tj][0] * inv; pv.y = sv[tj][1] * inv; pv.z = sv[tj][2] * inv; pv.w = sv[tj][3] * inv;
    if (a.P != nullptr) *reinterpret_cast<float4*>(a.P + o) = pv;      // (nullptr: the column partner saves the probabilities -- QUAD)
    pd[tj][0] = pv.x * seq_dmul(dk, key, a.pidx + o);
    pd[tj][1] = pv.y * seq_dmul(dk, key, a.pidx + o + 1);
    pd[tj][2] = pv.z * seq_dmul(dk, key, a.pidx + o + 2);
    pd[tj][3] = pv.w * seq_dmul(dk, key, a.pidx + o + 3);
  }
  f32x4 o[NQ];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct) {
    o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) o[ct] = GT_MFMA16(pd[tj][c], vb[ct][tj][c], o[ct]);
  }
  float* orow = ctx + (16 * ti + 4 * g) * ldc + l16;
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { if (!PAD || 16 * ct + l16 < hdr) orow[r * ldc + 16 * ct] = o[ct][r]; }
}

// The P values a wave's two backward roles read (its query tile's rows as 16-byte runs, its key tile's columns), for a caller that
// requests them ahead of the attention backward: P was written a forward phase ago through another L2 and is the coldest operand of the
// stage (SPLIT phases: requested before the state tiles, seq_attn_p_load).
struct SeqPPre { float4 pv[2]; float pc[8]; };
__device__ __forceinline__ SeqPPre seq_attn_p_load(const float* P, const int w, const int lane) {
  SeqPPre r;
  const int l16 = lane & 15, g = lane >> 4, i = 16 * w + l16;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) r.pv[tj] = *reinterpret_cast<const float4*>(P + (unsigned)(i * 32 + 16 * tj + 4 * g));
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int q = 0; q < 4; ++q) r.pc[4 * ti + q] = P[(unsigned)((16 * ti + 4 * g + q) * 32 + i)];
  return r;
}
// Backward, two roles per wave with a workgroup barrier between them (gt_attn.h): role 1 (query tile w) -> dq in registers and
// the row sums rd -> srd (32 floats of LDS per head); role 2 (key tile w) -> dk, dv in registers; after another barrier
// seq_attn_bwd_store writes dq / dk / dv over q / k / v of the head (the dqkv tile IS the qkv tile).
template <int HD, bool PAD>
__device__ __forceinline__ void seq_attn_bwd1(const SeqAttn& a, const float* dctx, const int lddc, const SeqDropK& dk, const uint32_t key,
                                              const int w, const int lane, float* srd, f32x4 (&dq_out)[HD / 16],
                                              const bool have_p = false, const SeqPPre& pp = SeqPPre()) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  const int i = 16 * w + l16;
  const float* dorow = dctx + i * lddc + 4 * g;
  const float* vrow = a.v + l16 * a.ldq + 4 * g;
  float4 df[NQ], v0[NQ], v1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    df[q] = seq_ld4<PAD>(dorow + 16 * q, 16 * q + 4 * g, hdr);
    v0[q] = seq_ld4<PAD>(vrow + 16 * q, 16 * q + 4 * g, hdr);
    v1[q] = seq_ld4<PAD>(vrow + 16 * a.ldq + 16 * q, 16 * q + 4 * g, hdr);
  }
  const float* kcol = a.k + 4 * g * a.ldq + l16;
  float kb[NQ][2][4];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) kb[ct][tj][c] = seq_ld1<PAD>(kcol + (16 * tj + c) * a.ldq + 16 * ct, 16 * ct + l16, hdr);
  float4 pv[2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) { if (have_p) pv[tj] = pp.pv[tj]; else pv[tj] = *reinterpret_cast<const float4*>(a.P + (unsigned)(i * 32 + 16 * tj + 4 * g)); }
  f32x4 dt[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};           // dPd^T tiles [tj]
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
    const uint32_t idx0 = a.pidx + (uint32_t)(i * 32 + 16 * tj + 4 * g);
    p[tj][0] = pv[tj].x; p[tj][1] = pv[tj].y; p[tj][2] = pv[tj].z; p[tj][3] = pv[tj].w;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dp[tj][r] = dt[tj][r] * seq_dmul(dk, key, idx0 + r);
      rd += dp[tj][r] * p[tj][r];
    }
  }
  rd += __shfl_xor(rd, 16); rd += __shfl_xor(rd, 32);
  if (g == 0) srd[i] = rd;
  float ds[2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[tj][r] = p[tj][r] * (dp[tj][r] - rd) * a.scale;
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct) {
    dq_out[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) dq_out[ct] = GT_MFMA16(ds[tj][c], kb[ct][tj][c], dq_out[ct]);
  }
}
template <int HD, bool PAD>
__device__ __forceinline__ void seq_attn_bwd2(const SeqAttn& a, const float* dctx, const int lddc, const SeqDropK& dk, const uint32_t key,
                                              const int w, const int lane, const float* srd, f32x4 (&dk_out)[HD / 16], f32x4 (&dv_out)[HD / 16],
                                              const bool have_p = false, const SeqPPre& pp = SeqPPre()) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? a.hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  const int j = 16 * w + l16;
  const float* dorow = dctx + l16 * lddc + 4 * g;                 // query tile 0; tile 1 = + 16 rows
  const float* vrow = a.v + j * a.ldq + 4 * g;
  float4 vf[NQ], d0[NQ], d1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    vf[q] = seq_ld4<PAD>(vrow + 16 * q, 16 * q + 4 * g, hdr);
    d0[q] = seq_ld4<PAD>(dorow + 16 * q, 16 * q + 4 * g, hdr);
    d1[q] = seq_ld4<PAD>(dorow + 16 * lddc + 16 * q, 16 * q + 4 * g, hdr);
  }
  const float* docol = dctx + 4 * g * lddc + l16;
  const float* qcol = a.q + 4 * g * a.ldq + l16;
  float db[NQ][2][4], qb[NQ][2][4];
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        db[ct][ti][c] = seq_ld1<PAD>(docol + (16 * ti + c) * lddc + 16 * ct, 16 * ct + l16, hdr);
        qb[ct][ti][c] = seq_ld1<PAD>(qcol + (16 * ti + c) * a.ldq + 16 * ct, 16 * ct + l16, hdr);
      }
  float pvv[2][4], rdv[2][4];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ti + 4 * g + r;
      pvv[ti][r] = have_p ? pp.pc[4 * ti + r] : a.P[(unsigned)(i * 32 + j)];
      rdv[ti][r] = srd[i];
    }
  f32x4 dd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};           // dPd tiles [ti]
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
      const float mk = seq_dmul(dk, key, a.pidx + (uint32_t)(i * 32 + j));
      pdm[ti][r] = pvv[ti][r] * mk;
      ds[ti][r] = pvv[ti][r] * (dd[ti][r] * mk - rdv[ti][r]) * a.scale;
    }
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct) {
    dv_out[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dk_out[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dv_out[ct] = GT_MFMA16(pdm[ti][c], db[ct][ti][c], dv_out[ct]);
        dk_out[ct] = GT_MFMA16(ds[ti][c], qb[ct][ti][c], dk_out[ct]);
      }
  }
}
// rows 16 w + 4 g + r, columns 16 ct + l16 of the head: dq over q, dk over k (+ d columns), dv over v (+ 2 d)
template <int HD, bool PAD>
__device__ __forceinline__ void seq_attn_bwd_store(float* dq, const int ldq, const int d, const int hd, const int w, const int lane,
                                                   const f32x4 (&dq_out)[HD / 16], const f32x4 (&dk_out)[HD / 16], const f32x4 (&dv_out)[HD / 16]) {
  constexpr int NQ = HD / 16;
  const int hdr = PAD ? hd : HD;
  const int l16 = lane & 15, g = lane >> 4;
  float* dqrow = dq + (16 * w + 4 * g) * ldq + l16;
#pragma unroll
  for (int ct = 0; ct < NQ; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (PAD && 16 * ct + l16 >= hdr) continue;
      dqrow[r * ldq + 16 * ct] = dq_out[ct][r];
      dqrow[r * ldq + 16 * ct + d] = dk_out[ct][r];
      dqrow[r * ldq + 16 * ct + 2 * d] = dv_out[ct][r];
    }
}

// ---- head_dim 2 (the reference's ClosedHH YAML: d_model 32, 16 heads) on the vector ALU.  Zero-padded to a 16-wide MFMA contraction such a head does
// 2 .. 8 useful multiplications per 16 and the stage is one latency chain per round of heads.  Here a (query row, head) pair's 32 scores,
// the softmax and the head's ctx columns live in the registers of one or two threads, operands straight from the LDS qkv tile as 8- /
// 16-byte reads (a wave's lanes differ in the head: consecutive addresses; in the row: broadcast).  Same P / dropout index layout as the
// MFMA form (P[head][query][key], index = head base + 32 query + key).  SPLIT kernels only (16 query / key rows per workgroup).
// (Written for head_dim 2 / 4 / 8; only 2 is instantiated: at 8 the compiler keeps 256 registers live and spills, and the
// testing YAML's head_dim-8 attention stays on the zero-padded MFMA form with every other head_dim below 16.)
template <int HD> struct SeqHv { float v[HD]; };
template <int HD>
__device__ __forceinline__ SeqHv<HD> seq_hv_ld(const float* p) {
  SeqHv<HD> r;
  if (HD == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
  else {
#pragma unroll
    for (int c = 0; c < HD; c += 4) { const float4 t = *reinterpret_cast<const float4*>(p + c); r.v[c] = t.x; r.v[c + 1] = t.y; r.v[c + 2] = t.z; r.v[c + 3] = t.w; }
  }
  return r;
}
template <int HD>
__device__ __forceinline__ float seq_hv_dot(const SeqHv<HD>& a, const SeqHv<HD>& b) {
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) t += a.v[c] * b.v[c];
  return t;
}
// the value of the neighbouring lane (lane ^ 1): one quad-permute DPP move
__device__ __forceinline__ float seq_lane_swap1(const float v) {
#ifdef GT_EMU
  return __shfl_xor(v, 1);
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
#endif
}
__device__ __forceinline__ uint32_t seq_lane_swap1u(const uint32_t v) { return __builtin_bit_cast(uint32_t, seq_lane_swap1(__builtin_bit_cast(float, v))); }
// Forward: TWO threads per (query row, head) -- neighbouring lanes, 16 keys each (16 rows x 16 heads of a SPLIT workgroup are 256 pairs:
// every thread of the workgroup busy, two waves per SIMD to hide each other's LDS and exp latency; one thread per pair with 32 keys
// left half of the SIMDs' issue slots empty: 18 k cycles per layer at the ClosedHH YAML shape).  The pair meets three times through a DPP
// move: row maximum, row sum, the ctx partial sums.  Lane `part` takes keys 0..15 in order, its neighbour 20..31, 16..19: the rotation
// puts the pair's 8-byte reads of one instruction 20 rows = 32 banks apart in the [32][3 d + 8] tile of d_model 32 (16 rows apart is the
// same bank).  2 x nrow x H <= the workgroup's threads.
template <int HD>
__device__ __forceinline__ void seq_attn_fwd_small(const float* sQ, const int ldq, const int d, const int H, const float scale, float* Pseq,
                                                   const uint32_t pidx_seq, float* sCtx, const int ldc, const SeqDropK& dk, const uint32_t key,
                                                   uint32_t* amask, const int row0, const int nrow, const int tid) {
  if (tid >= 2 * nrow * H) return;
  const int part = tid & 1, pr = tid >> 1, h = pr % H, i = row0 + pr / H;
  const int j0 = part ? 20 : 0, wrap = part ? 16 : 0;         // key of step jj: j0 + jj, minus wrap from jj = 12 on
  const float* ka = sQ + d + h * HD + j0 * ldq;
  const float* kb = ka - wrap * ldq;
  const SeqHv<HD> q = seq_hv_ld<HD>(sQ + i * ldq + h * HD);
  float sc[16], mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    sc[jj] = seq_hv_dot<HD>(q, seq_hv_ld<HD>((jj < 12 ? ka : kb) + jj * ldq)) * scale; mx = fmaxf(mx, sc[jj]);
    if ((jj & 3) == 3) { GT_SCHED_FENCE() }                    // (four keys' reads in flight)
  }
  mx = fmaxf(mx, seq_lane_swap1(mx));
  float sum = 0.f;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    sc[jj] = expf(sc[jj] - mx); sum += sc[jj];
    if ((jj & 3) == 3) { GT_SCHED_FENCE() }                    // (16 interleaved exp expansions are 80 registers of temporaries)
  }
  sum += seq_lane_swap1(sum);                                  // (a + b in both lanes: identical)
  const float inv = 1.0f / sum;
  float* Prow = Pseq + (size_t)h * 1024 + i * 32;
  const uint32_t pidx = pidx_seq + (uint32_t)(h * 1024 + i * 32);
  SeqHv<HD> o;
  uint32_t mb = 0u;
#pragma unroll
  for (int c = 0; c < HD; ++c) o.v[c] = 0.f;
#pragma unroll
  for (int jj = 0; jj < 16; jj += 4) {
    const int j = j0 + jj - (jj < 12 ? 0 : wrap);
    const float* vp = (jj < 12 ? ka : kb) + d + jj * ldq;
    float4 pv;
    pv.x = sc[jj] * inv; pv.y = sc[jj + 1] * inv; pv.z = sc[jj + 2] * inv; pv.w = sc[jj + 3] * inv;
    *reinterpret_cast<float4*>(Prow + j) = pv;
    float m0 = pv.x, m1 = pv.y, m2 = pv.z, m3 = pv.w;
    if (dk.thr) {                                                // (uniform) the keep decisions also go to the mask word: bit = step here
      const bool k0 = seq_dkeep(dk, key, pidx + j), k1 = seq_dkeep(dk, key, pidx + j + 1);
      const bool k2 = seq_dkeep(dk, key, pidx + j + 2), k3 = seq_dkeep(dk, key, pidx + j + 3);
      mb |= (k0 ? 1u << jj : 0u) | (k1 ? 2u << jj : 0u) | (k2 ? 4u << jj : 0u) | (k3 ? 8u << jj : 0u);
      m0 = k0 ? m0 * dk.scale : 0.f; m1 = k1 ? m1 * dk.scale : 0.f; m2 = k2 ? m2 * dk.scale : 0.f; m3 = k3 ? m3 * dk.scale : 0.f;
    }
    const SeqHv<HD> v0 = seq_hv_ld<HD>(vp), v1 = seq_hv_ld<HD>(vp + ldq), v2 = seq_hv_ld<HD>(vp + 2 * ldq), v3 = seq_hv_ld<HD>(vp + 3 * ldq);
#pragma unroll
    for (int c = 0; c < HD; ++c) o.v[c] += m0 * v0.v[c] + m1 * v1.v[c] + m2 * v2.v[c] + m3 * v3.v[c];
    GT_SCHED_FENCE()
  }
#pragma unroll
  for (int c = 0; c < HD; ++c) o.v[c] += seq_lane_swap1(o.v[c]);
  if (part == 0) {
#pragma unroll
    for (int c = 0; c < HD; ++c) sCtx[i * ldc + h * HD + c] = o.v[c];
  }
  // the row's 32 keep bits (bit = key) -> amask[head][query]: the backward reads them instead of hashing 32 + 16 indices per thread again
  // (three quarter-rate integer multiplications each: the hash was half of the attention backward's cycles at the ClosedHH YAML shape)
  if (dk.thr) {
    uint32_t w = part ? ((((mb << 4) | (mb >> 12)) & 0xFFFFu) << 16) : mb;      // lane 1: step jj is key 16 + (jj + 4) % 16
    w |= seq_lane_swap1u(w);
    if (part == 0) amask[h * 32 + i] = w;
  }
}
// Backward, two passes with a workgroup barrier between them (and one after: the results replace q / k / v in place; the barriers are
// the caller's).  Pass A, thread = (query row i of all 32, head): dP, the row sum rd -> srd[head][i], dS, dq_i; its P row and keep bits
// (SeqPRow) are REQUESTED by seq_attn_bwd_small_load, which the caller places at the start of the phase, ahead of the state tiles: P
// was written a forward phase ago by another XCD and takes the longest round trip of the launch.  Pass B, TWO threads per (key row j of
// the rows [krow0, krow0 + nkrow), head) -- neighbouring lanes, 16 queries each, summed through a DPP move: dv_j, dk_j over all 32
// queries; its P column and keep bits come from LDS (pass A's row image of P -- 64 KB: the FFN tile's place, idle here -- and the mask
// words behind the row sums): no global load between the two passes.  32 x H and 2 x nkrow x H <= the workgroup's threads; srd:
// 2 x H x 32 words of LDS.  The dropout mask of P comes from the forward's keep bits (amask[head][query], bit = key): no hash here.
struct SeqPRow { f32x4 p[8]; uint32_t mw; };
// 16 heads (d_model 32).  A thread's P row is 128 contiguous bytes, 4 KB from its neighbour lane's: read row-wise a wave instruction
// touches 64 lines for 1 KB of use (measured: the 64 KB of a sequence's P took 7.8 k cycles to arrive).  So the wave's 64 rows -- 16 heads x
// 4 queries, sixteen 512-byte runs -- are requested run-wise: instruction u brings heads 2 u and 2 u + 1, lane = (head & 1, query & 3,
// 16-byte slot); seq_attn_bwd_small_a turns them into rows through 8 KB of wave-private LDS.
__device__ __forceinline__ SeqPRow seq_attn_bwd_small_load(const float* Pseq, const uint32_t* amask, const int tid) {
  SeqPRow r;
  const int lane = tid & 63, w = tid >> 6;
  const float* src = Pseq + (size_t)(lane >> 5) * 1024 + (4 * w + ((lane >> 3) & 3)) * 32 + (lane & 7) * 4;
#pragma unroll
  for (int u = 0; u < 8; ++u) r.p[u] = *reinterpret_cast<const f32x4*>(src + (size_t)u * 2048);
  r.mw = amask[(tid & 15) * 32 + (tid >> 4)];
  return r;
}
template <int HD> struct SeqAttnSmallG { SeqHv<HD> dq, dkk, dvv; };
template <int HD>
__device__ __forceinline__ void seq_attn_bwd_small_a(SeqAttnSmallG<HD>& G, const float* sQ, const int ldq, const int d, const int H, const float scale,
                                                     const SeqPRow& pr, float* sImg, const float* sDO, const int lddo, const SeqDropK& dk, float* srd,
                                                     const int tid) {
#pragma unroll
  for (int c = 0; c < HD; ++c) { G.dq.v[c] = 0.f; G.dkk.v[c] = 0.f; G.dvv.v[c] = 0.f; }
  if (tid >= 32 * H) return;
  const int ha = tid % H, ia = tid / H;                        // (query row, head), head fastest
  const float msc = dk.thr ? dk.scale : 1.0f;
  const float* kp = sQ + d + ha * HD;
  const float* vp = sQ + 2 * d + ha * HD;
  const uint32_t mw = pr.mw | (dk.thr ? 0u : 0xFFFFFFFFu);
  const SeqHv<HD> dO = seq_hv_ld<HD>(sDO + ia * lddo + ha * HD);
  float p[32], ds[32];
  {
    // run-wise granules -> rows, through the wave's 8 KB of sImg: row (head h, query 4 w + il) lives at row slot 4 h + (il ^ (h >> 3)), its
    // 16-byte slot s at s ^ (h & 7): the 16 heads of one il -- a wave's lanes differ in the head first -- read 16 different (half, slot)
    // places of the 256-byte bank row.  LDS is in order within a wave: no barrier.
    const int lane = tid & 63;
    float* img = sImg + (tid >> 6) * 2048;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int h = 2 * u + (lane >> 5), il = (lane >> 3) & 3;
      *reinterpret_cast<f32x4*>(img + (4 * h + (il ^ (h >> 3))) * 32 + 4 * ((lane & 7) ^ (h & 7))) = pr.p[u];
    }
    GT_WAVE_SYNC();
    const int il = lane >> 4;
    const float* row = img + (4 * ha + (il ^ (ha >> 3))) * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
      const float4 t = *reinterpret_cast<const float4*>(row + 4 * ((j >> 2) ^ (ha & 7)));
      p[j] = t.x; p[j + 1] = t.y; p[j + 2] = t.z; p[j + 3] = t.w;
    }
  }
  float rd = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    ds[j] = seq_hv_dot<HD>(dO, seq_hv_ld<HD>(vp + j * ldq)) * (((mw >> j) & 1u) ? msc : 0.f);   // dP (under the probabilities' dropout mask; a select, no branch)
    rd += ds[j] * p[j];
    if ((j & 3) == 3) { GT_SCHED_FENCE() }
  }
  srd[ha * 32 + ia] = rd;
  reinterpret_cast<uint32_t*>(srd)[H * 32 + ha * 32 + ia] = mw;   // pass B's keep bits (behind the row sums)
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float g = p[j] * (ds[j] - rd) * scale;
    const SeqHv<HD> kj = seq_hv_ld<HD>(kp + j * ldq);
#pragma unroll
    for (int c = 0; c < HD; ++c) G.dq.v[c] += g * kj.v[c];
    if ((j & 3) == 3) { GT_SCHED_FENCE() }
  }
}
template <int HD>
__device__ __forceinline__ void seq_attn_bwd_small_b(SeqAttnSmallG<HD>& G, const float* sQ, const int ldq, const int d, const int H, const float scale,
                                                     const float* sImg, const float* sDO, const int lddo, const SeqDropK& dk,
                                                     const float* srd, const int krow0, const int nkrow, const int tid) {
  if (tid >= 2 * nkrow * H) return;
  const int pb = tid & 1, jb = krow0 + (tid >> 1) % nkrow, hb = (tid >> 1) / nkrow;        // (query half, key row, head)
  const float msc = dk.thr ? dk.scale : 1.0f;
  const int i0 = 16 * pb;                                      // this lane's queries: i0 .. i0 + 15
  const float* qp = sQ + hb * HD + i0 * ldq;
  const float* dop = sDO + hb * HD + i0 * lddo;
  const float* rdp = srd + hb * 32 + i0;
  const uint32_t* mwp = reinterpret_cast<const uint32_t*>(srd) + H * 32 + hb * 32 + i0;
  // P(hb, i, jb) from pass A's LDS image (seq_attn_bwd_small_a: wave i / 4, row slot 4 h + ((i & 3) ^ (h >> 3)), 16-byte slot ^ (h & 7))
  const float* pim = sImg + (4 * pb) * 2048 + (4 * hb) * 32 + 4 * ((jb >> 2) ^ (hb & 7)) + (jb & 3);
  const int hx = hb >> 3;
  const SeqHv<HD> v = seq_hv_ld<HD>(sQ + jb * ldq + 2 * d + hb * HD);
  const uint32_t allk = dk.thr ? 0u : 1u;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float pi = pim[(i >> 2) * 2048 + ((i & 3) ^ hx) * 32];
    const float mk = (((mwp[i] >> jb) | allk) & 1u) ? msc : 0.f;
    const SeqHv<HD> dO = seq_hv_ld<HD>(dop + i * lddo), qi = seq_hv_ld<HD>(qp + i * ldq);
    const float pm = pi * mk, g = pi * (seq_hv_dot<HD>(dO, v) * mk - rdp[i]) * scale;
#pragma unroll
    for (int c = 0; c < HD; ++c) { G.dvv.v[c] += pm * dO.v[c]; G.dkk.v[c] += g * qi.v[c]; }
    if ((i & 3) == 3) { GT_SCHED_FENCE() }
  }
#pragma unroll
  for (int c = 0; c < HD; ++c) { G.dvv.v[c] += seq_lane_swap1(G.dvv.v[c]); G.dkk.v[c] += seq_lane_swap1(G.dkk.v[c]); }
}
// every read of q / k / v is done (caller's barrier): dq / dk / dv go over them
template <int HD>
__device__ __forceinline__ void seq_attn_bwd_small_store(const SeqAttnSmallG<HD>& G, float* sQ, const int ldq, const int d, const int H,
                                                         const int krow0, const int nkrow, const int tid) {
  if (tid < 32 * H) {
    const int ha = tid % H, ia = tid / H;
#pragma unroll
    for (int c = 0; c < HD; ++c) sQ[ia * ldq + ha * HD + c] = G.dq.v[c];
  }
  if (tid < 2 * nkrow * H) {                                    // lane 0 of the pair stores dk_j, lane 1 dv_j
    const int pb = tid & 1, jb = krow0 + (tid >> 1) % nkrow, hb = (tid >> 1) / nkrow;
    float* dst = sQ + jb * ldq + (pb ? 2 * d : d) + hb * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) dst[c] = pb ? G.dvv.v[c] : G.dkk.v[c];
  }
}
#ifndef GT_SEQ_PPRE
#define GT_SEQ_PPRE 0           /* 1: SPLIT backward phases request the MFMA attention backward's P values ahead of the state tiles (measured: slower, tools/rejected/README.md) */
#endif
#ifndef GT_SEQ_VATTN
#define GT_SEQ_VATTN 1          /* 0: head_dim < 16 stays on the zero-padded MFMA form */
#endif

// SPLIT / QUAD kernels: which (sequence, part) a block is.  Blocks are dealt round-robin over the 8 XCDs (observed, speed only), so with
// the plain order the 2 / 4 workgroups of one sequence -- which all load the same q / k / v tile at the start of a phase, and in QUAD
// swap partial tiles -- sit on different L2s.  GT_SEQ_XCD_MAP: block x -> XCD x % 8, slot x / 8; the sequence's workgroups take
// consecutive slots of one XCD (its state crosses the fabric once per launch, not 2 / 4 times).  Any batch: nper x 8 x ceil(B / 8) is
// not the grid size, so the map applies when B % 8 == 0 and the plain order otherwise.  Returns sequence * nper + part.
#ifndef GT_SEQ_XCD_MAP
#define GT_SEQ_XCD_MAP 1
#endif
__device__ __forceinline__ int seq_vblock(const int bid, const int nper, const int B) {
  if (!GT_SEQ_XCD_MAP || (B & 7) != 0 || bid >= nper * B) return bid;
  const int xcd = bid & 7, slot = bid >> 3;
  return ((slot / nper) * 8 + xcd) * nper + slot % nper;
}

// ================================================================================================================ forward
// LDS geometry of a d_model class.  RP: most split-K parts a [32][d] result can come in (d = 16 -> 8 parts ... d > 64 -> 1).
template <int DP> struct SeqGeo {
  static constexpr int SX = DP + 8, SH = GT_SEQ_FMAX + 8, SQ = 3 * DP + 8, SRS = DP + 8, RP = DP == 32 ? 8 : DP == 64 ? 2 : 1, CW = DP / 16;
  static constexpr int TILE = 32 * SX, QKV = 32 * SQ, FFN = 32 * SH, RES = RP * 32 * SRS, NK = DP / 16;
  static constexpr int UNI = QKV > FFN ? QKV : FFN;          // the qkv tile and the FFN tile are never live together in the forward
};
// EXACT: d_model == DP (every shipped configuration): d is a compile-time constant, the matmul bodies are branch-free.
// SPLIT: TWO workgroups per sequence, 16 token rows each, and one launch per PHASE (a.phase):  every stage but attention is
// row-local, and attention needs the other half's K / V rows -- so a phase ends where those are produced (the in-proj) and the next
// launch picks its state up from the saved-activation buffers the backward needs anyway.  phase 0: input layer + in-proj(0);
// phase l + 1: attention(l) .. norm2(l), then in-proj(l + 1) or the output layer.  2 x batch workgroups fill twice the CUs: the
// path for small batches of the d_model-128 class, where a sequence's matmuls are bound by the MFMA rate of one CU.
// (Round 3: phase 0 and phase 1 are ONE launch -- the first phase computes input layer + in-proj(0) for all 32 rows; a.phase = layer.)
// QUAD (round 4; SPLIT, d_model 128, while 4 x batch workgroups fit the CUs): FOUR workgroups per sequence -- the two 16-row halves x two
// COLUMN PARTNERS (blockIdx.x & 1).  At batch 64 the SPLIT forward left 128 of the 256 CUs dark; the partners of a row half share its
// FFN, two thirds of a layer's matmul time: each computes HALF of dim_feedforward -- its 256 columns of FFN1, the partial FFN2 product
// over them -- and the two [16][128] partial results meet through ONE pair exchange per layer (seq_xchg_*: tagged 8-byte granules, one
// fabric round trip); both then run the (cheap) norm2 on identical sums.  The in-proj that closes the phase is split by columns too (its
// result goes to global memory for the next launch anyway: no exchange).  Attention, out-proj and norm1 -- a sixth of the phase -- are
// computed by both; saved activations are written by one of the two.  The last phase's output layer + loss run on partner 0 only.
// Both partners must be resident at once to meet: they are adjacent blocks of a grid that fits the chip (one workgroup per CU by LDS);
// the spin is bounded (seq_xchg_get).
// (The body is a device function over ONE LDS buffer -- lds, SeqFwdLds<DP>::N floats -- so that a kernel can run it and then the
//  backward's body on the same storage.  Returns false when this workgroup is done with the launch.)
template <int DP> struct SeqFwdLds { static constexpr int N = 3 * SeqGeo<DP>::TILE + SeqGeo<DP>::RES + SeqGeo<DP>::UNI; };
template <int DP, int HDC, bool EXACT, bool SPLIT, bool QUAD = false>
__device__ __forceinline__ bool seq_fwd_body(const SeqArgs& a, float* const lds) {
  static_assert(!QUAD || (SPLIT && EXACT && DP == 128), "QUAD: the SPLIT kernels of d_model 128");
  using G = SeqGeo<DP>;
  constexpr int SX = G::SX, SH = G::SH, SQ = G::SQ, SRS = G::SRS, CW = G::CW, NK = G::NK;
  constexpr int HD = SeqHd<HDC>::HD;
  constexpr bool PAD = SeqHd<HDC>::PAD;
  constexpr bool HALF = SPLIT;
  constexpr int NH = HALF ? 1 : 2, NROW = HALF ? 16 : 32;    // 16-row halves a matmul epilogue sees / rows this workgroup owns
  float* const sX = lds; float* const sX1 = sX + G::TILE; float* const sC = sX1 + G::TILE; float* const sR = sC + G::TILE; float* const sU = sR + G::RES;
  float* const sQ = sU;                                      // qkv tile: in-proj -> attention
  float* const sH = sU;                                      // FFN tile: FFN1 -> FFN2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  // (sequence, part) of this block.  Measured at the headline shape: the map takes the SPLIT backward phases from 103.7 to 101.5 us per step;
  // the QUAD forward LOSES 3 us with its four workgroups on one XCD (78.1 -> 81.1 us: they read the same lines of one L2 at the same
  // moments, and a write-through granule leaves the L2 it was written through; only the two ROW halves of a (sequence, partner) on one
  // XCD, partners on neighbouring ones: 0.2026 -> 0.2035 ms), so QUAD keeps the plain order
  const int vb = (SPLIT && !QUAD) ? seq_vblock((int)blockIdx.x, 2, a.B) : (int)blockIdx.x;
  const int cpart = QUAD ? (vb & 1) : 0;                     // QUAD: column partner
  const int half_id = QUAD ? (vb >> 1) : vb;                 // SPLIT: (sequence, row half)
  const int b = SPLIT ? half_id >> 1 : (int)blockIdx.x, rb = SPLIT ? 16 * (half_id & 1) : 0;
  const bool sv0 = !QUAD || cpart == 0, sv1 = !QUAD || cpart == 1;            // which partner saves what (each activation once)
  const int d = EXACT ? DP : a.d, F = a.F;
  const size_t r0 = (size_t)b * 32;                          // first token row of this sequence
  const float* const zp = gt_zero_ptr();
  const float* prm = a.prm;
  float* ws = a.ws;
  const SeqDropK dk = seq_dropk(a);
  const uint32_t idxd = (uint32_t)(r0 * d), idxf = (uint32_t)(r0 * F);   // dropout element index of this sequence's first row
  const float ascale = 1.0f / sqrtf((float)a.hd);
  // d_model 32 in the SPLIT kernels (the ClosedHH YAMLs: 16 workgroups x 2 on a 256-CU chip, every stage one dependent chain): a stage's global
  // operands -- weight fragments, bias / gamma / beta -- are requested one stage AHEAD, so that a phase pays the L2 round trip (1.8-2 k cycles)
  // once at its head instead of once per stage; the fragments are 8 ... 48 registers here (64 per stage at d_model 128, where the same was
  // measured slower in round 3).  Round 6.
  constexpr bool PF32 = SPLIT && !QUAD && EXACT && (DP == 32 || (DP == 64 && GT_SEQ_PF64)) && GT_SEQ_PF32;
  constexpr bool PFLN = PF32 || (SPLIT && EXACT && DP == 128 && GT_SEQ_PFLN128);     // the LayerNorm parameters alone: at d_model 128 too (24 registers per norm)
  SeqB<NK> ipre = SeqB<NK>();                                 // the next layer's in-proj fragment + bias chunk (requested in layer_rest)
  float4 ipre_b = make_float4(0.f, 0.f, 0.f, 0.f);
  bool have_ipre = false;
  // rows rb .. rb + nrows - 1 of a [32][ncol] tile of this sequence, global -> LDS, 16 bytes per thread and pass
  auto load_rows = [&](float* dst, const int str, const float* src, const int ncol, const int rb_, const int nrows) {
    const int q4 = ncol >> 2;
    for (int e = tid; e < nrows * q4; e += GT_SEQ_NT) {
      const int r = rb_ + e / q4, c = (e % q4) * 4;
      *reinterpret_cast<float4*>(dst + r * str + c) = *reinterpret_cast<const float4*>(src + (unsigned)(r * ncol + c));
    }
  };

  // ---- input layer: a0 = x Win^T + b; x0 = drop(relu(a0) + pe)     (A tile: the own input rows, zero-padded to 32 columns)
  // (all_tag: std::true_type = ALL 32 rows of the sequence even in a SPLIT workgroup -- the first SPLIT phase computes the input layer and
  //  layer 0's in-proj for the partner's rows too instead of waiting a launch for them: the weight fragments, which bound these stages,
  //  are streamed once either way)
  auto input_layer = [&](auto all_tag) {
    constexpr bool HF = HALF && !decltype(all_tag)::value;    // 16 own rows, or all 32
    constexpr int NHX = HF ? 1 : 2, NRX = HF ? 16 : 32;
    const int rbx = HF ? rb : 0;
    for (int e = tid; e < NRX * 32; e += GT_SEQ_NT) {
      const int r = rbx + (e >> 5), c = e & 31;
      sC[r * SX + c] = *(c < a.S ? a.xin + (r0 + r) * a.S + c : zp);
    }
    GT_BARRIER();
    GT_STAMP(1);
    const uint32_t key = seq_key(dk, GT_SITE_PE_ENC);
    float* ga0 = ws + a.a0 + r0 * d;
    float* gx0 = ws + a.x0 + r0 * d;
    seq_mm_edge<false, false, 2, HF>(sC + rbx * SX, SX, a.S, prm + a.in_w, a.S, d, prm + a.in_b, wave, lane, zp,
                                     [&](int n0, const f32x4& c0, const f32x4& c1, const float (&bi)[4]) {
#pragma unroll
      for (int h2 = 0; h2 < NHX; ++h2) {
        const int row = rbx + 16 * h2 + l16, col = n0 + 4 * lg;
        const f32x4& c = h2 ? c1 : c0;
        const float4 pe = *reinterpret_cast<const float4*>(a.pe + row * d + col);
        const unsigned o = (unsigned)(row * d + col);
        const float4 pre = make_float4(c[0] + bi[0], c[1] + bi[1], c[2] + bi[2], c[3] + bi[3]);
        float4 v;
        v.x = (fmaxf(pre.x, 0.f) + pe.x) * seq_dmul(dk, key, idxd + o);
        v.y = (fmaxf(pre.y, 0.f) + pe.y) * seq_dmul(dk, key, idxd + o + 1);
        v.z = (fmaxf(pre.z, 0.f) + pe.z) * seq_dmul(dk, key, idxd + o + 2);
        v.w = (fmaxf(pre.w, 0.f) + pe.w) * seq_dmul(dk, key, idxd + o + 3);
        if ((!HALF || HF || (row >= rb && row < rb + 16)) && sv0) {     // saved for the backward: every row once (its owner)
          *reinterpret_cast<float4*>(ga0 + o) = pre;
          *reinterpret_cast<float4*>(gx0 + o) = v;
        }
        *reinterpret_cast<float4*>(&sX[row * SX + col]) = v;
      }
    });
    GT_BARRIER();
  };
  // ---- in-proj of layer l: qkv = x Win^T + b -> the LDS qkv tile (own rows); ends with a barrier
  auto in_proj = [&](const int l, auto all_tag) {
    constexpr bool HF = HALF && !decltype(all_tag)::value;
    const int rbx = HF ? rb : 0;
    const float* pl = prm + (int64_t)l * a.pstride;
    const float* kf = ws + a.pack_f + (int64_t)l * a.kstride;
    GT_STAMP(2 + 10 * l);
    if constexpr (QUAD && HF) {
      // own rows, this partner's half of the 3 d columns (its 12 of the 24 column tiles): the tile leaves for global memory right after
      const int qc0 = cpart * (3 * d / 2);
      seq_mm_tiles<NK, (3 * DP / 32 + 7) / 8, EXACT, true>(sX + rbx * SX, SX, d, kf + (size_t)(qc0 >> 4) * NK * 256, 3 * d / 2, pl + a.p0.in_b + qc0,
                                                           wave, lane, [&](int n0, const f32x4& c0, const f32x4&, const float4& bi) {
        *reinterpret_cast<float4*>(&sQ[(rbx + l16) * SQ + qc0 + n0 + 4 * lg]) = make_float4(c0[0] + bi.x, c0[1] + bi.y, c0[2] + bi.z, c0[3] + bi.w);
      });
    } else if constexpr (HALF && !HF && EXACT) {
      // a SPLIT / QUAD workgroup computing layer 0's in-proj for the whole sequence (first phase): k and v for all 32 rows -- the attention
      // needs them -- but q for the OWN rows only (the other half's queries are its own business: an eighth of the MFMAs less;
      // headline 0.2021 -> 0.2002 ms)
      seq_mm_tiles<NK, (DP / 16 + 7) / 8, EXACT, true>(sX + rb * SX, SX, d, kf, d, pl + a.p0.in_b, wave, lane,
                                                       [&](int n0, const f32x4& c0, const f32x4&, const float4& bi) {
        *reinterpret_cast<float4*>(&sQ[(rb + l16) * SQ + n0 + 4 * lg]) = make_float4(c0[0] + bi.x, c0[1] + bi.y, c0[2] + bi.z, c0[3] + bi.w);
      });
      seq_mm_tiles<NK, (2 * DP / 16 + 7) / 8, EXACT, false>(sX, SX, d, kf + (size_t)(d >> 4) * NK * 256, 2 * d, pl + a.p0.in_b + d, wave, lane,
                                                            [&](int n0, const f32x4& c0, const f32x4& c1, const float4& bi) {
        const int col = d + n0 + 4 * lg;
        *reinterpret_cast<float4*>(&sQ[l16 * SQ + col]) = make_float4(c0[0] + bi.x, c0[1] + bi.y, c0[2] + bi.z, c0[3] + bi.w);
        *reinterpret_cast<float4*>(&sQ[(16 + l16) * SQ + col]) = make_float4(c1[0] + bi.x, c1[1] + bi.y, c1[2] + bi.z, c1[3] + bi.w);
      });
    } else {
      seq_mm_tiles<NK, (3 * DP / 16 + 7) / 8, EXACT, HF>(sX + rbx * SX, SX, d, kf, 3 * d, pl + a.p0.in_b, wave, lane,
                                                          [&](int n0, const f32x4& c0, const f32x4& c1, const float4& bi) {
        const int col = n0 + 4 * lg;
        *reinterpret_cast<float4*>(&sQ[(rbx + l16) * SQ + col]) = make_float4(c0[0] + bi.x, c0[1] + bi.y, c0[2] + bi.z, c0[3] + bi.w);
        if (!HF) *reinterpret_cast<float4*>(&sQ[(16 + l16) * SQ + col]) = make_float4(c1[0] + bi.x, c1[1] + bi.y, c1[2] + bi.z, c1[3] + bi.w);
      }, PF32 && HF && have_ipre, ipre, &ipre_b);
      have_ipre = false;
    }
    GT_BARRIER();
    GT_STAMP(2 + 10 * l + 1);
  };
  // ---- the rest of layer l: attention .. norm2 -> the next layer's input in sX (own rows); ends with a barrier
  // (returns true when this workgroup is done with the launch: QUAD's partner 1 in the last layer, once its FFN2 partial is on its way)
  auto layer_rest = [&](const int l, const bool save_qkv) -> bool {
    const float* pl = prm + (int64_t)l * a.pstride;          // this layer's parameters / saved activations (wave-uniform bases)
    const float* kf = ws + a.pack_f + (int64_t)l * a.kstride;                    // its fragment-ordered weights: in_w, out_w, w1, w2
    const float* kf_out = kf + 3 * d * d, *kf_w1 = kf + 4 * d * d, *kf_w2 = kf_w1 + d * F;
    float* wl = ws + (int64_t)l * a.wstride;
    const int site0 = GT_SITE_LAYER0 + 8 * l;
    const int sb = 2 + 10 * l;                               // stamp base of this layer
    // (QUAD, last layer of a launch that goes on into backward phase 0: each partner reads back ITS OWN saves there -- the LayerNorm
    //  statistics that otherwise one of the two writes are written by both, identical values)
    const bool fzl = QUAD && a.fuse_b0 != 0 && l + 1 == a.L;
    // ---- attention: operands from the LDS qkv tile, P to global, ctx to the LDS tile.  Whole: wave pair p takes head h4 + p and the
    // pair's two waves the two query tiles; SPLIT: wave w takes head h8 + w, query tile = the own half.  The qkv tile goes to global
    // here (saved for the backward) -- line-shaped, see seq_tile_out.
    if (save_qkv && sv0) seq_tile_out(wl + a.w0.qkv + r0 * 3 * d, sQ, SQ, 3 * d, tid, rb, NROW);
    SeqLnPre<CW> ln1p, ln2p;                                  // (PF32) the two norms' gamma / beta and the bias in front of them
    if constexpr (PFLN) seq_ln_pre<CW>(ln1p, pl + a.p0.out_b, pl + a.p0.n1w, pl + a.p0.n1b, tid);
#ifndef GT_SEQ_NO_PRE2
    const bool preo = SPLIT && EXACT && (DP > 64 || PF32);    // the out-proj's fragment: in flight under the attention (which loads nothing)
    SeqB<NK> bopre = SeqB<NK>();
    if (preo) bopre = seq_tiles_first<NK>(kf_out, d, d, wave, lane);
#else
    const bool preo = false;
    const SeqB<NK> bopre = SeqB<NK>();
#endif
    bool vattn = false;
    if constexpr (PAD && SPLIT && GT_SEQ_VATTN && (DP == 32 || DP == 64)) {        // (SPLIT kernels only: in the whole-sequence kernels the extra live range spills)
      vattn = a.hd == DP / 16 && a.H == 16;                   // (16 heads of 2 at d_model 32, of 4 at d_model 64: the backward's P staging is written for 16 heads)
      if (vattn) {
        const uint32_t key = seq_key(dk, site0 + GT_SITE_ATTN);
        float* Pseq = wl + a.w0.P + (size_t)(b * a.H) * 1024;
        const uint32_t pseq = (uint32_t)(b * a.H * 1024);
        seq_attn_fwd_small<DP / 16>(sQ, SQ, d, a.H, ascale, Pseq, pseq, sC, SX, dk, key, reinterpret_cast<uint32_t*>(ws + a.amask + (int64_t)l * a.amask_stride) + b * a.H * 32, rb, NROW, tid);
      }
    }
    if (!vattn) {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_ATTN);
      constexpr int HPR = HALF ? GT_SEQ_WAVES : GT_SEQ_WAVES / 2;              // heads per round
      for (int h4 = 0; h4 < a.H; h4 += HPR) {
        const int h = h4 + (HALF ? wave : wave >> 1);
        if (h < a.H) {
          SeqAttn at;
          at.q = sQ + h * a.hd; at.k = at.q + d; at.v = at.q + 2 * d; at.ldq = SQ; at.hd = a.hd; at.scale = ascale;
          at.pidx = (uint32_t)((b * a.H + h) * 1024); at.P = (sv0 && !(GT_SEQ_ACCT & 1)) ? wl + a.w0.P + (size_t)(b * a.H + h) * 1024 : nullptr;
          seq_attn_fwd<HD, PAD>(at, sC + h * a.hd, SX, dk, key, HALF ? (rb >> 4) : (wave & 1), lane);
        }
      }
    }
    GT_BARRIER();
    GT_STAMP(sb + 2);
    // ---- out-proj (raw product -> sR part 0); the ctx tile also goes to global here (operand of the out-proj weight gradient)
    // (PF32) ALL of FFN1's fragments + bias chunks: in flight under the out-proj and norm1.  (Requested ahead of the attention instead they
    //  arrive under its P stores -- 106 KB per workgroup through the same 64 B/clk vector-memory path: the attention stage took 1.7 k cycles longer.)
    constexpr int F1T = GT_SEQ_FMAX / (QUAD ? 256 : 128);    // FFN1 tiles per wave
    SeqTilesAll<NK, F1T> f1p;
    if constexpr (PF32) seq_tiles_all_load<NK, F1T>(f1p, kf_w1, d, F, pl + a.p0.b1, wave, lane);
    {
      if (sv1 && !(GT_SEQ_ACCT & 1)) seq_tile_out(wl + a.w0.ctx + r0 * d, sC, SX, d, tid, rb, NROW);
      if (DP <= 64 && !SPLIT) seq_mm_square(sC, SX, d, kf_out, sR, SRS, wave, lane);
      else
        seq_mm_tiles<NK, 1, EXACT, HALF>(sC + rb * SX, SX, d, kf_out, d, nullptr, wave, lane, [&](int n0, const f32x4& c0, const f32x4& c1, const float4&) {
          *reinterpret_cast<float4*>(&sR[(rb + l16) * SRS + n0 + 4 * lg]) = make_float4(c0[0], c0[1], c0[2], c0[3]);
          if (!HALF) *reinterpret_cast<float4*>(&sR[(16 + l16) * SRS + n0 + 4 * lg]) = make_float4(c1[0], c1[1], c1[2], c1[3]);
        }, preo, bopre);
    }
    GT_BARRIER();
    GT_STAMP(sb + 3);
    // ---- z1 = drop(. + b_o) + x;  norm1 -> x1
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROP1);
      const float* bo = pl + a.p0.out_b;
      seq_ln_fwd<DP, HALF>([&](int row, int c0, float (&z)[CW]) {
        float bi[CW], xr[CW];
        SeqVec<CW>::ld(z, &sR[row * SRS + c0]); SeqVec<CW>::ld(xr, &sX[row * SX + c0]);
        if constexpr (PFLN) {
#pragma unroll
          for (int e = 0; e < CW; ++e) bi[e] = ln1p.bi[e];
        } else SeqVec<CW>::ld(bi, bo + c0);
#pragma unroll
        for (int e = 0; e < CW; ++e) z[e] = (z[e] + bi[e]) * seq_dmul(dk, key, idxd + (uint32_t)(row * d + c0 + e)) + xr[e];
      }, sX1, SX, d, pl + a.p0.n1w, pl + a.p0.n1b, ((sv1 || fzl) && !(GT_SEQ_ACCT & 1)) ? wl + a.w0.x1 + r0 * d : nullptr, wl + a.w0.xhat1 + r0 * d, wl + a.w0.rstd1 + r0, tid, rb,
         PFLN ? &ln1p : nullptr);
    }
    GT_BARRIER();
    GT_STAMP(sb + 4);
    // ---- FFN1: hact = drop(relu(x1 W1^T + b1))     (QUAD: this partner's half of the columns, fc0 .. fc0 + F / 2)
    const int fc0 = QUAD ? cpart * (F >> 1) : 0, fcn = QUAD ? F >> 1 : F;
    if constexpr (PFLN) seq_ln_pre<CW>(ln2p, pl + a.p0.b2, pl + a.p0.n2w, pl + a.p0.n2b, tid);     // norm2's parameters ...
    if constexpr (PF32) {                                        // ... and the NEXT layer's in-proj fragment: in flight under FFN1 and FFN2
      if (l + 1 < a.L) {
        const float* kfn = ws + a.pack_f + (int64_t)(l + 1) * a.kstride;
        const int tcl = wave < (3 * d >> 4) ? wave : 0;
        seq_b_load<NK, true>(ipre, kfn, d >> 4, tcl, 0, d >> 4, lane);
        ipre_b = *reinterpret_cast<const float4*>(prm + (int64_t)(l + 1) * a.pstride + a.p0.in_b + 16 * tcl + 4 * lg);
        have_ipre = true;
      }
    }
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_FFN);
      GT_SUBSET(l == 1);                                         // (diagnostic builds: sub-stage stamps of this stage, wave 0)
      auto ffn1_epi = [&](int n0, const f32x4& c0, const f32x4& c1, const float4& bi) {
        const int col = fc0 + n0 + 4 * lg;
#pragma unroll
        for (int h2 = 0; h2 < NH; ++h2) {
          const int row = rb + 16 * h2 + l16;
          const f32x4& c = h2 ? c1 : c0;
          const unsigned o = (unsigned)(row * F + col);
          float4 v;
          v.x = fmaxf(c[0] + bi.x, 0.f) * seq_dmul(dk, key, idxf + o);
          v.y = fmaxf(c[1] + bi.y, 0.f) * seq_dmul(dk, key, idxf + o + 1);
          v.z = fmaxf(c[2] + bi.z, 0.f) * seq_dmul(dk, key, idxf + o + 2);
          v.w = fmaxf(c[3] + bi.w, 0.f) * seq_dmul(dk, key, idxf + o + 3);
          *reinterpret_cast<float4*>(&sH[row * SH + col]) = v;
        }
      };
      if constexpr (PF32) seq_mm_tiles_all<NK, F1T, HALF>(sX1 + rb * SX, SX, d, fcn, wave, lane, f1p, ffn1_epi);
      else seq_mm_tiles<NK, F1T, EXACT, HALF>(sX1 + rb * SX, SX, d, kf_w1 + (size_t)(fc0 >> 4) * (d >> 4) * 256, fcn, pl + a.p0.b1 + fc0, wave, lane, ffn1_epi);
    }
    const int nkf = F >> 4, kq0 = QUAD ? cpart * (nkf >> 1) : 0;            // QUAD: this partner's k-steps of FFN2, kq0 .. kq0 + nkf / 2
#ifndef GT_SEQ_NO_PRE
    const bool pre2 = QUAD ? ((nkf >> 1) & 7) == 0 : (SPLIT && seq_splitk_pre_ok(F, d));
    SeqB<8> b2pre = SeqB<8>();
    if (pre2) {                                                      // FFN2's first chunk: in flight across the barrier and the tile store
      if (QUAD) seq_b_load<8, true>(b2pre, kf_w2, nkf, wave, kq0, 8, lane);
      else b2pre = seq_splitk_first(kf_w2, F, d, wave, lane);
    }
#else
    const bool pre2 = false;
    const SeqB<8> b2pre = SeqB<8>();
#endif
    GT_BARRIER();
    GT_SUBSET(false);
    GT_STAMP(sb + 5);
    // ---- FFN2 (K = F: split over the waves) -> partial tiles; the FFN tile goes to global (saved for the backward)
    if constexpr (QUAD) {
      // this partner's K half, one column tile per wave; its partial tile goes to the partner (and to sR part `cpart`), the partner's
      // arrives as part 1 - cpart: both sum part 0 + part 1, in that order, and continue on identical values
      if (!(GT_SEQ_ACCT & 1)) seq_tile_out_cols(wl + a.w0.hact + r0 * F, F, sH, SH, fc0, fcn, tid, rb, NROW);
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      seq_mm_krange<true>(acc0, acc1, sH + rb * SH + l16 * SH + 4 * lg, SH, kf_w2, nkf, wave, kq0, kq0 + (nkf >> 1), lane, pre2, b2pre);
      GT_STAMP(300 + 4 * l);
      unsigned long long* const xq = reinterpret_cast<unsigned long long*>(ws + a.xchg) + 8;       // (granule 0..7: the region's header)
      const bool last = l + 1 == a.L;                               // last layer: partner 1 only sends (partner 0 runs the output layer alone)
      const bool fz = a.fuse_b0 != 0;                               // (the launch goes on into backward phase 0: partner 1 needs the sum too)
      if (!last || cpart == 1 || fz) seq_xchg_put(xq + (size_t)vb * GT_XCHG_WG_GRANULES, acc0, tid);
      if (last && cpart == 1 && !fz) return true;                   // (workgroup-uniform)
      *reinterpret_cast<float4*>(&sR[(16 * cpart + l16) * SRS + 16 * wave + 4 * lg]) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
      GT_STAMP(301 + 4 * l);
      const f32x4 oth = seq_xchg_get(xq + (size_t)(vb ^ 1) * GT_XCHG_WG_GRANULES, tid, reinterpret_cast<unsigned*>(ws + a.xchg), a.spin_max);
      GT_STAMP(302 + 4 * l);
      *reinterpret_cast<float4*>(&sR[(16 * (1 - cpart) + l16) * SRS + 16 * wave + 4 * lg]) = make_float4(oth[0], oth[1], oth[2], oth[3]);
    } else {
      seq_tile_out(wl + a.w0.hact + r0 * F, sH, SH, F, tid, rb, NROW);
      seq_mm_splitk<HALF>(sH + rb * SH, SH, F, kf_w2, d, sR + rb * SRS, SRS, wave, lane, pre2, b2pre);
    }
    GT_BARRIER();
    GT_STAMP(sb + 6);
    // ---- z2 = drop(sum of the parts + b2) + x1;  norm2 -> the next layer's input
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROPF);
      const float* b2 = pl + a.p0.b2;
      const int parts = seq_splitk_parts(d);
      seq_ln_fwd<DP, HALF>([&](int row, int c0, float (&z)[CW]) {
        float bi[CW], xr[CW];
        if constexpr (QUAD) {                                        // sR rows 0..15: partner 0's partial tile of the own rows, 16..31: partner 1's
          float u[CW];
          SeqVec<CW>::ld(z, &sR[(row - rb) * SRS + c0]); SeqVec<CW>::ld(u, &sR[(16 + row - rb) * SRS + c0]);
#pragma unroll
          for (int e = 0; e < CW; ++e) z[e] += u[e];
        } else {
          seq_parts_sum<CW>(z, sR, SRS, parts, row, c0);
        }
        SeqVec<CW>::ld(xr, &sX1[row * SX + c0]);
        if constexpr (PFLN) {
#pragma unroll
          for (int e = 0; e < CW; ++e) bi[e] = ln2p.bi[e];
        } else SeqVec<CW>::ld(bi, b2 + c0);
#pragma unroll
        for (int e = 0; e < CW; ++e) z[e] = (z[e] + bi[e]) * seq_dmul(dk, key, idxd + (uint32_t)(row * d + c0 + e)) + xr[e];
      }, sX, SX, d, pl + a.p0.n2w, pl + a.p0.n2b, (sv0 || fzl) ? wl + a.w0.xout + r0 * d : nullptr, wl + a.w0.xhat2 + r0 * d, wl + a.w0.rstd2 + r0, tid, rb,
         PFLN ? &ln2p : nullptr);
    }
    GT_BARRIER();
    GT_STAMP(sb + 7);
    return false;
  };
  // ---- final encoder norm -> memory, then the output layer: [h logits | sigmoid v | 0.5 tanh o]
  auto output_layer = [&]() {
    seq_ln_fwd<DP, HALF>([&](int row, int c0, float (&z)[CW]) { SeqVec<CW>::ld(z, &sX[row * SX + c0]); }, sC, SX, d, prm + a.encn_w, prm + a.encn_b,
                         ws + a.memory + r0 * d, ws + a.enc_xhat + r0 * d, ws + a.enc_rstd + r0, tid, rb);
    GT_BARRIER();
    seq_mm_edge<false, true, NK, HALF>(sC + rb * SX, SX, d, prm + a.out_w, d, GT_TGT, prm + a.out_b, wave, lane, zp,
                                       [&](int n0, const f32x4& c0, const f32x4& c1, const float (&bi)[4]) {
#pragma unroll
      for (int h2 = 0; h2 < NH; ++h2) {
        const f32x4& c = h2 ? c1 : c0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = n0 + 4 * lg + r;
          if (col < GT_TGT) {
            float v = c[r] + bi[r];
            if (col >= 2 * GT_VOICES) v = 0.5f * tanhf(v);
            else if (col >= GT_VOICES) v = gt_sigmoid(v);
            a.hvo[(r0 + rb + 16 * h2 + l16) * GT_TGT + col] = v;
            sR[(rb + 16 * h2 + l16) * 28 + col] = v;          // (sR is free here: the loss tail reads the [32][28] output tile from it)
          }
        }
      }
    });
    if (a.loss_y == nullptr) return;
    const unsigned lwg_n = QUAD ? gridDim.x >> 1 : gridDim.x, lwg_id = QUAD ? (unsigned)half_id : blockIdx.x;   // workgroups that run the output layer
    // ---- fused loss: one thread per (own row, voice); partial sums of this workgroup -> loss_part[blockIdx][4]; the last workgroup
    // to arrive (ticket) adds all partials in a fixed order -> bitwise-reproducible statistics, as in loss_kernel
    GT_BARRIER();
    float* red = sX1;                                         // [8 waves][4] + flag (the x1 tile is dead)
    const float invM = 1.0f / (float)(a.B * 32);
    float bce = 0.f, mv = 0.f, mo = 0.f, ok = 0.f;
    if (tid < NROW * GT_VOICES) {
      const int row = rb + tid / GT_VOICES, j = tid % GT_VOICES;
      const size_t base = (r0 + row) * GT_TGT + j;
      float gh, gv, go;
      gt_loss_elem<true>(sR[row * 28 + j], sR[row * 28 + j + GT_VOICES], sR[row * 28 + j + 2 * GT_VOICES], a.loss_y[base],
                         a.loss_y[base + GT_VOICES], a.loss_y[base + 2 * GT_VOICES], a.loss_penalty, invM, bce, mv, mo, ok, gh, gv, go);
      ws[a.dlogits + base] = gh; ws[a.dlogits + base + GT_VOICES] = gv; ws[a.dlogits + base + 2 * GT_VOICES] = go;
    }
    bce = gt_wave_sum(bce); mv = gt_wave_sum(mv); mo = gt_wave_sum(mo); ok = gt_wave_sum(ok);
    if (lane == 0) { red[wave * 4 + 0] = bce; red[wave * 4 + 1] = mv; red[wave * 4 + 2] = mo; red[wave * 4 + 3] = ok; }
    GT_BARRIER();
    if (tid == 0 && cpart != 0) red[32] = 0.0f;                // (QUAD, fused with backward phase 0: partner 1 computed dlogits for itself; the statistics are partner 0's)
    if (tid == 0 && cpart == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = red[q];
        for (int w = 1; w < GT_SEQ_WAVES; ++w) t += red[w * 4 + q];
        gt_pub_store(a.loss_part + lwg_id * 4 + q, t);
      }
      // the four partials are write-through (agent-scope) stores, drained before the ticket; the last arriver -- told by the value its
      // add returns -- reads all partials with agent-scope loads: no __threadfence(), which would write back every dirty line of this
      // XCD's L2 (the activations just saved: microseconds, in every workgroup) -- MI355X_MICROARCH.md, valid hand-off forms, row 1
      const unsigned t = gt_pub_ticket(a.loss_ticket);
      red[32] = (t == lwg_n - 1) ? 1.0f : 0.0f;
    }
    GT_BARRIER();
    if (red[32] == 0.0f) return;
    if (wave < 4) {                                           // wave q sums quantity q: lane l takes workgroups l, l + 64, ..., then the xor tree
      float acc = 0.f;
      for (unsigned bk = lane; bk < lwg_n; bk += 64) acc += gt_pub_load(a.loss_part + bk * 4 + wave);
      acc = gt_wave_sum(acc);
      if (lane == 0) red[40 + wave] = acc * invM;
    }
    GT_BARRIER();
    if (tid == 0) {
      const float b_ = red[40], v_ = red[41], o_ = red[42];
      a.loss_stats[0] = b_ + v_ + o_;
      a.loss_stats[1] = red[43] * (1.0f / GT_VOICES);
      a.loss_stats[2] = 0.f;
      a.loss_stats[3] = b_; a.loss_stats[4] = v_; a.loss_stats[5] = o_; a.loss_stats[6] = 0.f; a.loss_stats[7] = 0.f;
      *a.loss_ticket = 0u;                                    // re-arm for the next step
    }
  };

#ifdef GT_SEQ_STAMPS
  GT_SUBSET(false);
  GT_BARRIER();
#endif
  typedef std::true_type AllRows;
  typedef std::false_type OwnRows;
  if (!SPLIT) {
    GT_STAMP(0);
    input_layer(OwnRows{});
    for (int l = 0; l < a.L; ++l) { in_proj(l, OwnRows{}); (void)layer_rest(l, true); }
    output_layer();
    GT_STAMP(2 + 10 * a.L);
  } else {
    // SPLIT phase l (one launch per encoder layer): [phase 0: input layer + in-proj(0) for ALL rows | else: the layer input's own rows and
    // the whole sequence's q / k / v from the workspace], attention .. norm2 of layer l on the own rows, then in-proj(l + 1) of the own
    // rows (saved: the next launch's operands) or the output layer + loss
    const int l = a.phase;
    if constexpr (QUAD) {
      // a.quad_pro: a PROLOGUE launch (a.phase < 0) computes the input layer and in-proj(0) once -- own rows, own column half, like every
      // later in-proj -- instead of all four workgroups of a sequence computing them for all 32 rows inside phase 0
      if (l < 0) {
        input_layer(OwnRows{});
        in_proj(0, OwnRows{});
        seq_tile_out_cols(ws + a.w0.qkv + r0 * 3 * d, 3 * d, sQ, SQ, cpart * (3 * d / 2), 3 * d / 2, tid, rb, NROW);
        return false;
      }
    }
    const bool pro = QUAD && a.quad_pro != 0;
    GT_STAMP(60 + 2 * (l + 1));
    GT_WGSTAMP(0);
    if (l == 0 && !pro) {
      input_layer(AllRows{});
      in_proj(0, AllRows{});
    } else {
      load_rows(sX, SX, ws + (l == 0 ? a.x0 : (int64_t)(l - 1) * a.wstride + a.w0.xout) + r0 * d, d, rb, NROW);   // own rows of the layer input
      // k / v of the whole sequence, q of the own rows (the other half's queries are its own business: a sixth of the bytes less)
      {
        const float* gq = ws + (int64_t)l * a.wstride + a.w0.qkv + r0 * 3 * d;
        const int q4k = (2 * d) >> 2, q4q = d >> 2;
        for (int e = tid; e < 32 * q4k; e += GT_SEQ_NT) {
          const int r = e / q4k, c = d + (e % q4k) * 4;
          *reinterpret_cast<float4*>(sQ + r * SQ + c) = *reinterpret_cast<const float4*>(gq + (unsigned)(r * 3 * d + c));
        }
        for (int e = tid; e < NROW * q4q; e += GT_SEQ_NT) {
          const int r = rb + e / q4q, c = (e % q4q) * 4;
          *reinterpret_cast<float4*>(sQ + r * SQ + c) = *reinterpret_cast<const float4*>(gq + (unsigned)(r * 3 * d + c));
        }
      }
      GT_BARRIER();
    }
    GT_STAMP(2 + 10 * l + 1);
    if (layer_rest(l, l == 0 && !pro)) return false;
    if (l + 1 < a.L) {
      in_proj(l + 1, OwnRows{});
      float* const gq = ws + (int64_t)(l + 1) * a.wstride + a.w0.qkv + r0 * 3 * d;
      if (QUAD) seq_tile_out_cols(gq, 3 * d, sQ, SQ, cpart * (3 * d / 2), 3 * d / 2, tid, rb, NROW);
      else seq_tile_out(gq, sQ, SQ, 3 * d, tid, rb, NROW);
    } else {
      output_layer();
    }
    GT_WGSTAMP(1);
    GT_STAMP(61 + 2 * (l + 1));
  }
  return true;
}
template <int DP, int HDC, bool EXACT, bool SPLIT, bool QUAD = false>
__global__ __launch_bounds__(GT_SEQ_NT) void seq_fwd_kernel(SeqArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[SeqFwdLds<DP>::N];
  (void)seq_fwd_body<DP, HDC, EXACT, SPLIT, QUAD>(a, lds);
}

// ================================================================================================================ backward
// LayerNorm jobs (dgamma / dbeta partial blocks, one [2][d] row per workgroup) in the order the kernel fills them: 0 = final norm, then
// for l = L-1 .. 0: 1 + 2 (L-1-l) = norm2 of layer l, 2 + 2 (L-1-l) = norm1 of layer l.  The host registers them in this order.
// SPLIT (two workgroups per sequence, 16 rows each, one launch per phase): only the attention backward couples the rows (dk / dv
// contract over all queries), so a phase ends at dctx -- own rows to the hand-over buffer a.dctx -- and the next one begins with the
// attention backward of the WHOLE sequence, computed by both workgroups of the pair (a fifth of a phase; no exchange, no
// second launch), each keeping its own rows of dq / dk / dv.  phase 0: output-layer dgrad .. out-proj dgrad of layer L-1;
// phase p: attention backward + in-proj dgrad of layer L-p, then norm2 backward .. out-proj dgrad of layer L-p-1 (or the input
// layer's backward).
// QUAD (round 4): backward phase 0 -- the one phase without riders, so half of the chip idles -- with four workgroups per sequence, as in
// the forward: the column partners of a row half split the FFN2 dgrad by columns of dhid and the FFN1 dgrad by its contraction (one
// pair exchange of the [16][128] partial results), split the out-proj dgrad's columns (dctx goes to the hand-over buffer anyway), and
// compute the output-layer dgrad and the LayerNorm backward passes twice; partner 0 writes what both computed.  Phase 0 only.
template <int DP> struct SeqBwdLds {
  using G = SeqGeo<DP>;
  static constexpr int U = DP > 64 ? G::UNI : G::QKV + G::FFN, P = 2 * GT_SEQ_WAVES * DP;
  static constexpr int N = 3 * G::TILE + G::RES + U + P + 32 * GT_SEQ_WAVES / 2;
};
template <int DP, int HDC, bool EXACT, bool SPLIT, bool QUAD = false>
__device__ __forceinline__ void seq_bwd_body(const SeqArgs& a, float* const lds) {
  static_assert(!QUAD || (SPLIT && EXACT && DP == 128), "QUAD: the SPLIT kernels of d_model 128");
  using G = SeqGeo<DP>;
  constexpr int SX = G::SX, SH = G::SH, SQ = G::SQ, SRS = G::SRS, CW = G::CW, NK = G::NK;
  constexpr int HD = SeqHd<HDC>::HD;
  constexpr bool PAD = SeqHd<HDC>::PAD;
  constexpr bool HALF = SPLIT;
  constexpr int NROW = HALF ? 16 : 32;
  // sZ: gradient w.r.t. the last layer's output (first LayerNorm backward only), then dctx; sDZ: the LayerNorm backward's dz (residual
  // gradient); sC: dz * dropout mask (A operand of the next dgrad); sQ: the saved qkv tile of the layer, overwritten IN PLACE by
  // dq / dk / dv in the attention backward; sH: the FFN tile (hact, then dhid in place).  ALIAS (DP 128: 160 KB do not hold both):
  // sQ and sH share storage -- the FFN tile is dead once the FFN1 dgrad has read it, the qkv tile is loaded after that, and the
  // next layer's FFN tile is requested only after the in-proj dgrad has read dqkv.
  constexpr bool ALIAS = DP > 64;
  float* const sZ = lds; float* const sDZ = sZ + G::TILE; float* const sC = sDZ + G::TILE; float* const sR = sC + G::TILE;
  float* const sU = sR + G::RES; float* const sP = sU + SeqBwdLds<DP>::U; float* const srd = sP + SeqBwdLds<DP>::P;
  float* const sQ = sU;
  float* const sH = ALIAS ? sU : sU + G::QKV;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const int vb = (SPLIT && !QUAD) ? seq_vblock((int)blockIdx.x, 2, a.B) : (int)blockIdx.x;      // (riders: blocks >= 2 B keep their index)
  const int cpart = QUAD ? (vb & 1) : 0, half_id = QUAD ? (vb >> 1) : vb;               // QUAD: column partner; (sequence, row half)
  const int b = SPLIT ? half_id >> 1 : (int)blockIdx.x, rb = SPLIT ? 16 * (half_id & 1) : 0;
  const bool sv0 = !QUAD || cpart == 0;                                                  // this workgroup writes what both partners compute
  const int d = EXACT ? DP : a.d, F = a.F;
  const size_t r0 = (size_t)b * 32;
  if constexpr (SPLIT && DP == 128 && !QUAD) {
    // rider workgroups (gt_seq_wg.h): the blocks behind the sequence workgroups compute the weight gradients whose operands the
    // earlier phases left in the workspace -- same kernel, same LDS footprint, hence always on a CU no sequence workgroup occupies
    if (a.grd != nullptr && (int)blockIdx.x >= a.nseq) {
      static_assert(SeqBwdLds<DP>::U >= GT_WG_LDS && SeqBwdLds<DP>::P >= 8 * 64, "rider LDS");
      seq_wg_riders(a, a.phase, (int)blockIdx.x - a.nseq, (int)gridDim.x - a.nseq, sU, sP, tid);
      return;
    }
  }
  const float* const zp = gt_zero_ptr();
  const float* prm = a.prm;
  float* ws = a.ws;
  const SeqDropK dk = seq_dropk(a);
  const uint32_t idxd = (uint32_t)(r0 * d);
  const float mscale = dk.thr ? dk.scale : 1.0f;
  const float ascale = 1.0f / sqrtf((float)a.hd);
  auto part_at = [&](int job) { return ws + a.ln_part + (int64_t)job * a.ln_part_stride + (size_t)(QUAD ? half_id : (int)blockIdx.x) * 2 * d; };
  // rows rb_ .. rb_ + nrows - 1 of a [32][ncol] tile of this sequence, global -> LDS, 16 bytes per thread and pass
  auto load_rows = [&](float* dst, const int str, const float* src, const int ncol, const int rb_, const int nrows) {
    const int q4 = ncol >> 2;
    for (int e = tid; e < nrows * q4; e += GT_SEQ_NT) {
      const int r = rb_ + e / q4, c = (e % q4) * 4;
      *reinterpret_cast<float4*>(dst + r * str + c) = *reinterpret_cast<const float4*>(src + (unsigned)(r * ncol + c));
    }
  };

  // ---- output layer dgrad: dmem = dlogits Wout (A tile: own rows x 27, zero-padded to 32 columns); final norm backward -> the
  // gradient w.r.t. the last layer's output (sZ, in place; its dz is not a weight-gradient operand)
  auto prologue = [&]() {
    for (int e = tid; e < NROW * 32; e += GT_SEQ_NT) {
      const int r = rb + (e >> 5), c = e & 31;
      sC[r * SX + c] = *(c < GT_TGT ? ws + a.dlogits + (r0 + r) * GT_TGT + c : zp);
    }
    if (!SPLIT) load_rows(sH, SH, ws + a.w0.hact + (int64_t)(a.L - 1) * a.wstride + r0 * F, F, 0, 32);
    GT_BARRIER();
    seq_mm_edge<true, true, 2, HALF>(sC + rb * SX, SX, GT_TGT, prm + a.out_w, d, d, nullptr, wave, lane, zp,
                                     [&](int n0, const f32x4& c0, const f32x4& c1, const float (&)[4]) {
      const int col = n0 + 4 * lg;
      *reinterpret_cast<float4*>(&sZ[(rb + l16) * SX + col]) = make_float4(c0[0], c0[1], c0[2], c0[3]);
      if (!HALF) *reinterpret_cast<float4*>(&sZ[(16 + l16) * SX + col]) = make_float4(c1[0], c1[1], c1[2], c1[3]);
    });
    GT_BARRIER();
    {
      SeqDropK nd = dk; nd.thr = 0u;
      seq_ln_bwd<DP, HALF>([&](int row, int c0, float (&g)[CW]) { SeqVec<CW>::ld(g, &sZ[row * SX + c0]); }, sZ, sC, SX, d, ws + a.enc_xhat + r0 * d,
                           ws + a.enc_rstd + r0, prm + a.encn_w, nd, 0u, 0u, nullptr, nullptr, sP, tid, rb);
    }
    GT_BARRIER();
    if (sv0) seq_ln_part<DP, HALF>(sP, part_at(0), d, tid);
    GT_BARRIER();                                                 // (sP is rewritten by the first norm2 backward below)
    GT_STAMP(101);
  };
  // d_model 32, SPLIT (round 6, as in the forward): the chain's saved LayerNorm operands are requested at the phase's start
  constexpr bool PFB = SPLIT && !QUAD && EXACT && (DP == 32 || (DP == 64 && GT_SEQ_PF64)) && GT_SEQ_PF32;
  constexpr bool PFBLN = PFB || (SPLIT && !QUAD && EXACT && DP == 128 && GT_SEQ_PFLN128);    // the LayerNorm operands alone: at d_model 128 too (17 registers per norm)
  SeqLnBwdPre<CW> lb2p, lb1p;
  bool have_lbp = false;                                     // (set by chain_prefetch, at the start of a phase > 0)
  constexpr int F2T = GT_SEQ_FMAX / (QUAD ? 256 : 128);      // FFN2 dgrad tiles per wave
  SeqTilesAll<NK, F2T> f2p;                                   // ALL of the FFN2 dgrad's fragments (requested at the chain's head)
  // ---- the row-local chain of layer l: norm2 backward .. out-proj dgrad -> dctx in sZ (own rows); ends with a barrier.
  // fromg: g = gradient w.r.t. this layer's output comes from sZ (layer L-1); else g = the in-proj dgrad parts of layer l+1 + its dz1
  auto chain = [&](const int l, const bool fromg) {
    const float* pl = prm + (int64_t)l * a.pstride;
    const float* kb = ws + a.pack_b + (int64_t)l * a.kstride;                    // dgrad-ordered weights: in_w, out_w, w1, w2
    const float* kb_out = kb + 3 * d * d, *kb_w1 = kb + 4 * d * d, *kb_w2 = kb_w1 + d * F;
    float* wl = ws + (int64_t)l * a.wstride;
    float* tl = ws + (int64_t)l * a.tstride;
    const int site0 = GT_SITE_LAYER0 + 8 * l, jb = 1 + 2 * (a.L - 1 - l);
    const int sb = 102 + 10 * (a.L - 1 - l);
    // ---- norm2 backward -> dz2 -> sDZ, dz2 * mask(dropout on the FFN output) -> sC; both to global for the weight gradients
    // the layer's hact tile (the FFN2 dgrad's mask, then dhid in place).  SPLIT: its 16 own rows are REQUESTED here -- at most four
    // 16-byte loads per thread, clamped addresses, no branch around a load -- and go to LDS only after the norm2 backward below: a
    // load -> LDS copy in front of it made every thread sit out the cold L2 round trip before the stage's own loads were even issued
    // (norm2 bwd 7.5 k cycles against norm1 bwd's 3.6 k)
    static_assert(16 * GT_SEQ_FMAX / 4 <= 4 * GT_SEQ_NT, "four 16-byte loads per thread cover the 16 x F tile");
    if constexpr (PFB) seq_tiles_all_load<NK, F2T, false>(f2p, kb_w2, d, F, nullptr, wave, lane);      // ALL of the FFN2 dgrad's fragments: in flight under the norm2 backward
    float4 hp0 = make_float4(0.f, 0.f, 0.f, 0.f), hp1 = hp0, hp2 = hp0, hp3 = hp0;     // (named: an indexed array went to scratch)
    // (QUAD: this partner's half of the columns, fc0 .. fc0 + F / 2)
    const int fc0 = QUAD ? cpart * (F >> 1) : 0, fcn = QUAD ? F >> 1 : F;
    const int hq4 = fcn >> 2, hn = 16 * hq4;
    auto hoff = [&](const int u) { const int e = tid + u * GT_SEQ_NT, ec = e < hn ? e : hn - 1; return (unsigned)((rb + ec / hq4) * F + fc0 + (ec % hq4) * 4); };
    if (SPLIT) {
      const float* src = wl + a.w0.hact + r0 * F;
      hp0 = *reinterpret_cast<const float4*>(src + hoff(0)); hp1 = *reinterpret_cast<const float4*>(src + hoff(1));
      hp2 = *reinterpret_cast<const float4*>(src + hoff(2)); hp3 = *reinterpret_cast<const float4*>(src + hoff(3));
    } else if (ALIAS && !fromg) {
      load_rows(sH, SH, wl + a.w0.hact + r0 * F, F, rb, NROW);   // (whole, !ALIAS: requested a layer ahead)
    }
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROPF);
      const int parts = seq_splitk_parts(d);
      seq_ln_bwd<DP, HALF>([&](int row, int c0, float (&g)[CW]) {
        if (fromg) SeqVec<CW>::ld(g, &sZ[row * SX + c0]);
        else {
          float r[CW];
          seq_parts_sum<CW>(g, sR, SRS, parts, row, c0); SeqVec<CW>::ld(r, &sDZ[row * SX + c0]);
#pragma unroll
          for (int e = 0; e < CW; ++e) g[e] += r[e];
        }
      }, sDZ, sC, SX, d, wl + a.w0.xhat2 + r0 * d, wl + a.w0.rstd2 + r0, pl + a.p0.n2w, dk, key, idxd, sv0 ? tl + a.t0.dzA + r0 * d : nullptr,
                           (dk.thr && sv0) ? tl + a.t0.dzAm + r0 * d : nullptr, sP, tid, rb, (PFBLN && have_lbp) ? &lb2p : nullptr);
    }
    if (SPLIT) {
      auto hst = [&](const int u, const float4& v) { const int e = tid + u * GT_SEQ_NT; if (e < hn) *reinterpret_cast<float4*>(sH + (rb + e / hq4) * SH + fc0 + (e % hq4) * 4) = v; };
      hst(0, hp0); hst(1, hp1); hst(2, hp2); hst(3, hp3);
    }
    GT_BARRIER();
    GT_STAMP(sb);
    // ---- FFN2 dgrad: dhid = (dz2m W2) * [hact != 0] * 1/(1-p), in place over the hact tile in sH
    if (sv0) seq_ln_part<DP, HALF>(sP, part_at(jb), d, tid);
    auto ffn2d_epi = [&](int n0, const f32x4& c0, const f32x4& c1, const float4&) {
      const int col = fc0 + n0 + 4 * lg;
#pragma unroll
      for (int h2 = 0; h2 < (HALF ? 1 : 2); ++h2) {
        float* hp = &sH[(rb + 16 * h2 + l16) * SH + col];
        const f32x4& c = h2 ? c1 : c0;
        const float4 ha = *reinterpret_cast<const float4*>(hp);
        *reinterpret_cast<float4*>(hp) = make_float4(ha.x != 0.f ? c[0] * mscale : 0.f, ha.y != 0.f ? c[1] * mscale : 0.f,
                                                     ha.z != 0.f ? c[2] * mscale : 0.f, ha.w != 0.f ? c[3] * mscale : 0.f);
      }
    };
    if constexpr (PFB) seq_mm_tiles_all<NK, F2T, HALF>(sC + rb * SX, SX, d, fcn, wave, lane, f2p, ffn2d_epi);
    else seq_mm_tiles<NK, F2T, EXACT, HALF>(sC + rb * SX, SX, d, kb_w2 + (size_t)(fc0 >> 4) * (d >> 4) * 256, fcn, nullptr, wave, lane, ffn2d_epi);
    const int nkf = F >> 4, kq0 = QUAD ? cpart * (nkf >> 1) : 0;            // QUAD: this partner's k-steps of the FFN1 dgrad
#ifndef GT_SEQ_NO_PRE
    const bool pre1 = QUAD ? ((nkf >> 1) & 7) == 0 : (SPLIT && seq_splitk_pre_ok(F, d));
    SeqB<8> b1pre = SeqB<8>();
    if (pre1) {                                                      // FFN1 dgrad's first chunk: in flight across the barrier and the tile store
      if (QUAD) seq_b_load<8, true>(b1pre, kb_w1, nkf, wave, kq0, 8, lane);
      else b1pre = seq_splitk_first(kb_w1, F, d, wave, lane);
    }
#else
    const bool pre1 = false;
    const SeqB<8> b1pre = SeqB<8>();
#endif
    GT_BARRIER();
    GT_STAMP(sb + 1);
    SeqB<NK> bodpre = SeqB<NK>();                             // (PFB) the out-proj dgrad's fragment: in flight under the FFN1 dgrad and the norm1 backward
    if constexpr (PFB) bodpre = seq_tiles_first<NK>(kb_out, d, d, wave, lane);
    // ---- FFN1 dgrad (K = F: split over the waves) -> partial tiles; dhid goes to global (operand of both FFN weight gradients)
    if constexpr (QUAD) {
      // the own half of the contraction, one column tile per wave; the partial tiles are swapped with the partner (seq_xchg_*), both sum
      // part 0 + part 1 in that order (sR rows 0..15: partner 0's tile of the own rows, 16..31: partner 1's)
      seq_tile_out_cols(tl + a.t0.dhid + r0 * F, F, sH, SH, fc0, fcn, tid, rb, NROW);
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      seq_mm_krange<true>(acc0, acc1, sH + rb * SH + l16 * SH + 4 * lg, SH, kb_w1, nkf, wave, kq0, kq0 + (nkf >> 1), lane, pre1, b1pre);
      // (fused behind the last forward phase: a region of its own -- the forward's exchange of this launch may still be in flight at the partner)
      const int64_t xoff = a.fuse_b0 ? a.xchg_b : a.xchg;
      unsigned long long* const xq = reinterpret_cast<unsigned long long*>(ws + xoff) + 8;
      seq_xchg_put(xq + (size_t)vb * GT_XCHG_WG_GRANULES, acc0, tid);
      *reinterpret_cast<float4*>(&sR[(16 * cpart + l16) * SRS + 16 * wave + 4 * lg]) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
      const f32x4 oth = seq_xchg_get(xq + (size_t)(vb ^ 1) * GT_XCHG_WG_GRANULES, tid, reinterpret_cast<unsigned*>(ws + a.xchg), a.spin_max);       // (one error word: the forward region's header)
      *reinterpret_cast<float4*>(&sR[(16 * (1 - cpart) + l16) * SRS + 16 * wave + 4 * lg]) = make_float4(oth[0], oth[1], oth[2], oth[3]);
    } else {
      seq_tile_out(tl + a.t0.dhid + r0 * F, sH, SH, F, tid, rb, NROW);
      seq_mm_splitk<HALF>(sH + rb * SH, SH, F, kb_w1, d, sR + rb * SRS, SRS, wave, lane, pre1, b1pre);
    }
    GT_BARRIER();
    GT_STAMP(sb + 2);
    // ---- norm1 backward: g1 = parts + dz2 -> dz1 -> sDZ, dz1 * mask(dropout1) -> sC.  Whole: the saved qkv tile of this layer is
    // requested now (ALIAS: over the FFN tile, which the FFN1 dgrad has finished reading); SPLIT loads it at the next phase's start
    if (!SPLIT) load_rows(sQ, SQ, wl + a.w0.qkv + r0 * 3 * d, 3 * d, 0, 32);
    {
      const uint32_t key = seq_key(dk, site0 + GT_SITE_DROP1);
      const int parts = seq_splitk_parts(d);
      seq_ln_bwd<DP, HALF>([&](int row, int c0, float (&g)[CW]) {
        float r[CW];
        if constexpr (QUAD) {
          float u[CW];
          SeqVec<CW>::ld(g, &sR[(row - rb) * SRS + c0]); SeqVec<CW>::ld(u, &sR[(16 + row - rb) * SRS + c0]);
#pragma unroll
          for (int e = 0; e < CW; ++e) g[e] += u[e];
        } else {
          seq_parts_sum<CW>(g, sR, SRS, parts, row, c0);
        }
        SeqVec<CW>::ld(r, &sDZ[row * SX + c0]);
#pragma unroll
        for (int e = 0; e < CW; ++e) g[e] += r[e];
      }, sDZ, sC, SX, d, wl + a.w0.xhat1 + r0 * d, wl + a.w0.rstd1 + r0, pl + a.p0.n1w, dk, key, idxd, sv0 ? tl + a.t0.dzB + r0 * d : nullptr,
                           (dk.thr && sv0) ? tl + a.t0.dzBm + r0 * d : nullptr, sP, tid, rb, (PFBLN && have_lbp) ? &lb1p : nullptr);
    }
    GT_BARRIER();
    GT_STAMP(sb + 3);
    // ---- out-proj dgrad: dctx = dz1m Wo -> sZ (LDS: the attention backward reads it there)
    if (sv0) seq_ln_part<DP, HALF>(sP, part_at(jb + 1), d, tid);
    if (DP <= 64 && !SPLIT) seq_mm_square(sC, SX, d, kb_out, sZ, SX, wave, lane);
    else if constexpr (QUAD)        // this partner's half of the dctx columns (d / 32 tiles, one per wave): they leave for the hand-over buffer
      seq_mm_tiles<NK, 1, EXACT, true>(sC + rb * SX, SX, d, kb_out + (size_t)cpart * (d >> 5) * NK * 256, d >> 1, nullptr, wave, lane,
                                       [&](int n0, const f32x4& c0, const f32x4&, const float4&) {
        *reinterpret_cast<float4*>(&sZ[(rb + l16) * SX + cpart * (d >> 1) + n0 + 4 * lg]) = make_float4(c0[0], c0[1], c0[2], c0[3]);
      });
    else
      seq_mm_tiles<NK, 1, EXACT, HALF>(sC + rb * SX, SX, d, kb_out, d, nullptr, wave, lane, [&](int n0, const f32x4& c0, const f32x4& c1, const float4&) {
        *reinterpret_cast<float4*>(&sZ[(rb + l16) * SX + n0 + 4 * lg]) = make_float4(c0[0], c0[1], c0[2], c0[3]);
        if (!HALF) *reinterpret_cast<float4*>(&sZ[(16 + l16) * SX + n0 + 4 * lg]) = make_float4(c1[0], c1[1], c1[2], c1[3]);
      }, PFB, bodpre);
    GT_BARRIER();
    GT_STAMP(sb + 4);
  };
  // ---- attention backward of the whole sequence (four heads at a time: q / k / v from the LDS tile, P from global (saved), dctx
  // from LDS; dq / dk / dv replace q / k / v of the head in place -- every wave of the round has finished reading before anyone
  // stores: third barrier), own rows of dqkv -> global (operand of the in-proj weight gradient), in-proj dgrad (K = 3 d: split over
  // the waves) -> partial tiles; ends with a barrier
  // (PFB) the chain of layer l - 1 follows the attention backward + in-proj dgrad of layer l: its two norms' saved operands (x-hat, rstd, gamma: three
  // small loads per thread each) are requested at the phase's START.  Measured alternatives for these AND the chain's bigger operands (the own rows of
  // the hact tile, the FFN2 dgrad's fragments: 106 KB per workgroup): at the in-proj dgrad's head that stage waits for them (its own fragment loads are
  // branchy at K = 96: the compiler's vmcnt(0) covers everything in flight, + 2.4 k cycles); between the two passes of the attention backward the
  // second pass takes 2.8 k longer; the big ones at the phase's start sit under the attention's state and P loads (+ 1.3 k).  They stay at the chain's head.
  auto chain_prefetch = [&](const int l) {
    if constexpr (PFBLN) {
      if (l > 0) {
        const float* wp = ws + (int64_t)(l - 1) * a.wstride;
        const float* pp = prm + (int64_t)(l - 1) * a.pstride;
        seq_ln_bwd_pre<CW>(lb2p, d, wp + a.w0.xhat2 + r0 * d, wp + a.w0.rstd2 + r0, pp + a.p0.n2w, tid, rb);
        seq_ln_bwd_pre<CW>(lb1p, d, wp + a.w0.xhat1 + r0 * d, wp + a.w0.rstd1 + r0, pp + a.p0.n1w, tid, rb);
        have_lbp = true;
      }
    }
  };
  SeqPRow prow = SeqPRow();                                  // head_dim-2 attention: this thread's P row, requested at the start of the phase
  SeqPPre ppre = SeqPPre();                                  // MFMA attention: the wave's P values of the first round of heads, likewise
  bool have_ppre = false;
  auto attn_inproj = [&](const int l) {
    const float* kb = ws + a.pack_b + (int64_t)l * a.kstride;
    float* wl = ws + (int64_t)l * a.wstride;
    float* tl = ws + (int64_t)l * a.tstride;
    const int sb = 102 + 10 * (a.L - 1 - l);
#ifndef GT_SEQ_NO_PRE4
    const bool preq = SPLIT && seq_splitk_pre_ok(3 * d, d);
#else
    const bool preq = false;
#endif
    SeqB<8> bqpre = SeqB<8>();
    bool vattn = false;
    if constexpr (PAD && SPLIT && GT_SEQ_VATTN && (DP == 32 || DP == 64)) {        // (SPLIT kernels only: in the whole-sequence kernels the extra live range spills)
      vattn = a.hd == DP / 16 && a.H == 16;
      if (vattn) {                                              // (sR is free here: H x 32 row sums)
        static_assert(G::FFN >= 16 * 1024 && G::RES >= 2 * 16 * 32 && !ALIAS, "head_dim-2 attention backward: P image in the FFN tile, row sums + keep bits in sR");
        SeqAttnSmallG<DP / 16> G;
        seq_attn_bwd_small_a<DP / 16>(G, sQ, SQ, d, a.H, ascale, prow, sH, sZ, SX, dk, sR, tid);
        GT_BARRIER();
        GT_STAMP(400 + 4 * a.phase + 1);
        seq_attn_bwd_small_b<DP / 16>(G, sQ, SQ, d, a.H, ascale, sH, sZ, SX, dk, sR, rb, NROW, tid);
        GT_BARRIER();
        GT_STAMP(400 + 4 * a.phase + 2);
        seq_attn_bwd_small_store<DP / 16>(G, sQ, SQ, d, a.H, rb, NROW, tid);
#ifndef GT_SEQ_NO_PRE4
        if (preq) bqpre = seq_splitk_first(kb, 3 * d, d, wave, lane);
#endif
      }
    }
    if (!vattn) {
      const uint32_t key = seq_key(dk, GT_SITE_LAYER0 + 8 * l + GT_SITE_ATTN);
      for (int h4 = 0; h4 < a.H; h4 += GT_SEQ_WAVES / 2) {
        const int h = h4 + (wave >> 1);
        const bool active = h < a.H;
        SeqAttn at;
        at.q = sQ + h * a.hd; at.k = at.q + d; at.v = at.q + 2 * d; at.ldq = SQ; at.hd = a.hd; at.scale = ascale;
        at.pidx = (uint32_t)((b * a.H + h) * 1024); at.P = wl + a.w0.P + (size_t)(b * a.H + h) * 1024;
        f32x4 dq_out[HD / 16], dk_out[HD / 16], dv_out[HD / 16];
        const bool hp = have_ppre && h4 == 0;                          // (the first round's P values were requested at the start of the phase)
        if (active) seq_attn_bwd1<HD, PAD>(at, sZ + h * a.hd, SX, dk, key, wave & 1, lane, srd + 32 * (wave >> 1), dq_out, hp, ppre);
        GT_BARRIER();
        if (active) seq_attn_bwd2<HD, PAD>(at, sZ + h * a.hd, SX, dk, key, wave & 1, lane, srd + 32 * (wave >> 1), dk_out, dv_out, hp, ppre);
#ifndef GT_SEQ_NO_PRE4
        // the in-proj dgrad's first chunk: in flight across the two barriers, the dq / dk / dv store and the dqkv tile's way to global
        if (preq && h4 + GT_SEQ_WAVES / 2 >= a.H) bqpre = seq_splitk_first(kb, 3 * d, d, wave, lane);
#endif
        GT_BARRIER();
        if (active) seq_attn_bwd_store<HD, PAD>(sQ + h * a.hd, SQ, d, a.hd, wave & 1, lane, dq_out, dk_out, dv_out);
      }
    }
    GT_BARRIER();
    GT_STAMP(sb + 5);
    seq_tile_out(tl + a.t0.dqkv + r0 * 3 * d, sQ, SQ, 3 * d, tid, rb, NROW);
    if (!SPLIT && !ALIAS && l > 0) load_rows(sH, SH, ws + (int64_t)(l - 1) * a.wstride + a.w0.hact + r0 * F, F, 0, 32);   // the next layer's FFN tile
    seq_mm_splitk<HALF>(sQ + rb * SQ, SQ, 3 * d, kb, d, sR + rb * SRS, SRS, wave, lane, preq, bqpre);
    GT_BARRIER();
    GT_STAMP(sb + 6);
  };
  // ---- InputLayer backward: da0 = (parts + dz1) * dropout mask * [a0 > 0] -> global (operand of the input layer's weight gradient)
  auto input_bwd = [&]() {
    const uint32_t key = seq_key(dk, GT_SITE_PE_ENC);
    const int parts = seq_splitk_parts(d);
    const int row = rb + (tid >> 4), c0 = (tid & 15) * CW;
    if (c0 < d && (!HALF || tid < 256)) {
      const unsigned o = (unsigned)(row * d + c0);
      float g[CW], r[CW], a0v[CW];
      seq_parts_sum<CW>(g, sR, SRS, parts, row, c0); SeqVec<CW>::ld(r, &sDZ[row * SX + c0]); SeqVec<CW>::ld(a0v, ws + a.a0 + r0 * d + o);
#pragma unroll
      for (int e = 0; e < CW; ++e) { const float v = (g[e] + r[e]) * seq_dmul(dk, key, idxd + o + e); g[e] = a0v[e] > 0.f ? v : 0.f; }
      SeqVec<CW>::st(ws + a.da0 + r0 * d + o, g);
    }
  };

#ifdef GT_SEQ_STAMPS
  GT_SUBSET(false);
  GT_BARRIER();
#endif
  if constexpr (QUAD) {
    GT_STAMP(160);
    prologue();
    chain(a.L - 1, true);
    seq_tile_out_cols(ws + a.dctx + r0 * d, d, sZ, SX, cpart * (d >> 1), d >> 1, tid, rb, NROW);
    GT_STAMP(161);
  } else if (!SPLIT) {
    GT_STAMP(100);
    prologue();
    for (int l = a.L - 1; l >= 0; --l) { chain(l, l == a.L - 1); attn_inproj(l); }
    input_bwd();
    GT_STAMP(102 + 10 * a.L);
  } else if (a.phase == 0) {
    GT_STAMP(160);
    prologue();
    chain(a.L - 1, true);
    seq_tile_out(ws + a.dctx + r0 * d, sZ, SX, d, tid, rb, NROW);
    GT_STAMP(161);
  } else {
    const int l = a.L - a.phase;
    GT_STAMP(160 + 2 * a.phase);
    const int64_t hand = (int64_t)a.B * 32 * d;                                                             // floats per hand-over buffer
    bool vpre = false;
    if constexpr (PAD && SPLIT && GT_SEQ_VATTN && (DP == 32 || DP == 64)) vpre = a.hd == DP / 16 && a.H == 16;
    if (GT_SEQ_PPRE && !vpre && (wave >> 1) < a.H) {
      ppre = seq_attn_p_load(ws + (int64_t)l * a.wstride + a.w0.P + (size_t)(b * a.H + (wave >> 1)) * 1024, wave & 1, lane);
      have_ppre = true;
    }
    if constexpr (PAD && SPLIT && GT_SEQ_VATTN && (DP == 32 || DP == 64)) {
      if (a.hd == DP / 16 && a.H == 16)
        prow = seq_attn_bwd_small_load(ws + (int64_t)l * a.wstride + a.w0.P + (size_t)(b * a.H) * 1024,
                                       reinterpret_cast<const uint32_t*>(ws + a.amask + (int64_t)l * a.amask_stride) + b * a.H * 32, tid);
    }
    load_rows(sZ, SX, ws + a.dctx + ((a.phase - 1) & 1) * hand + r0 * d, d, 0, 32);                         // dctx of the whole sequence
    load_rows(sQ, SQ, ws + (int64_t)l * a.wstride + a.w0.qkv + r0 * 3 * d, 3 * d, 0, 32);                   // its saved q / k / v
    load_rows(sDZ, SX, ws + (int64_t)l * a.tstride + a.t0.dzB + r0 * d, d, rb, NROW);                        // dz1 of layer l, own rows
#if GT_SEQ_PFB_LN
    chain_prefetch(l);
#endif
    GT_BARRIER();
    GT_STAMP(400 + 4 * a.phase);
    attn_inproj(l);
    if (l > 0) {
      chain(l - 1, false);
      seq_tile_out(ws + a.dctx + (a.phase & 1) * hand + r0 * d, sZ, SX, d, tid, rb, NROW);
    } else {
      input_bwd();
    }
    GT_STAMP(161 + 2 * a.phase);
  }
}
template <int DP, int HDC, bool EXACT, bool SPLIT, bool QUAD = false>
__global__ __launch_bounds__(GT_SEQ_NT) void seq_bwd_kernel(SeqArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[SeqBwdLds<DP>::N];
  seq_bwd_body<DP, HDC, EXACT, SPLIT, QUAD>(a, lds);
}
// QUAD, fused train step: the LAST forward phase and backward phase 0 in one launch (a.fuse_b0).  From the loss down to the out-proj
// dgrad of the last layer the backward is row-local -- the same rows, the same (row half, column partner) workgroup as the forward phase
// that ends in the loss -- so the launch boundary between them bought nothing but its cost (kernel duration minus first-workgroup-start ..
// last-workgroup-end: 4 us) and phase 0's reload of what this workgroup had just written.  Both partners run the whole forward phase here
// (the last layer's pair exchange in both directions, the output layer and dlogits twice, the statistics once), then the backward body on
// the same LDS buffer: it reads its operands from global memory as ever -- this workgroup's own stores of a moment ago, ordered by the
// barrier below (workgroup scope: one CU, one L1).
template <int HDC>
__global__ __launch_bounds__(GT_SEQ_NT) void seq_fb_kernel(SeqArgs a) {
  constexpr int N = SeqBwdLds<128>::N > SeqFwdLds<128>::N ? SeqBwdLds<128>::N : SeqFwdLds<128>::N;
  __shared__ __attribute__((aligned(16))) float lds[N];
  if (!seq_fwd_body<128, HDC, true, true, true>(a, lds)) return;
  __syncthreads();
  SeqArgs b = a;
  b.phase = 0;
  seq_bwd_body<128, 32, true, true, true>(b, lds);
}
