// Common device/host helpers for libgroove_hip.so (gfx950 / CDNA4 only).
// GT_EMU is defined only by tests/emu/build_emu.sh (host fiber emulator, test infrastructure).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef GT_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
// D[row=(reg&3)+8*(reg>>2)+4*(l>>5)][col=l&31], reg in [0,16).  Exact f32 fmaf chain, the same rate as the 16x16x4 form.
#define GT_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15];
// D[row=4*(l>>4)+reg][col=l&15].  Exact f32 fmaf chain (cdna_hip_programming.md section 3).
#define GT_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// v_mfma_f32_16x16x32_bf16 (gt_config.precision = 1): lane l supplies A[i=l&15][k=8(l>>4)+j] and B[k=8(l>>4)+j][col l&15], j = 0..7
// (cdna_hip_programming.md 3, "A/B operand lane maps, bf16"); C/D as the f32 form above.  fp32 accumulate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define GT_MFMA16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// v_mfma_f32_32x32x16_bf16: lane l (r = l&31, h = l>>5) supplies A[row r][k = 8h + j] and B[k = 8h + j][col r]; C/D as the f32 32x32 form
#define GT_MFMA32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define GT_BF16X8_SET(vec, j, x) (vec)[j] = (__bf16)(x)
// f32 -> bf16, round to nearest even, NaN stays NaN: the plain cast lowers to v_cvt_pk_bf16_f32 (MI355X_MICROARCH.md,
// "Correctness boundaries")
__device__ __forceinline__ uint16_t gt_f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
#endif
#ifdef GT_EMU
static inline uint32_t gt_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float gt_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#else
__device__ __forceinline__ uint32_t gt_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float gt_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
#endif
__host__ __device__ static inline float gt_bf2f(uint16_t h) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)h << 16;
  return c.f;
}

#include "../../include/groove_hip.h"

// GT_BARRIER(): stage barrier of the sequence-resident kernels (gt_seq.h): waits for this wave's LDS traffic
// only -- __syncthreads() would also drain its vector-memory queue (vmcnt(0)), which prefetches and saves keep in flight
// across barriers (cdna_hip_programming.md 5, "Pipelining across barriers").
// (The LDS-DMA staging path this file once carried -- global_load_lds + inline-asm LDS reads -- measured ~27 GB/s per CU
// against 46-70 GB/s through registers and was removed; DESIGN.md 3, rejected experiments.)
#ifdef GT_EMU
#define GT_BARRIER() __syncthreads()
#else
#define GT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// GT_WAVE_SYNC(): lanes of one wave exchange data through LDS without a workgroup barrier (wave-private staging, gt_seq_wg.h).  The
// hardware runs a wave's lanes in lock-step and its LDS instructions in order, so this is only a compiler scheduling barrier; the host
// emulator's lanes are independent fibers and rendezvous here.
#ifdef GT_EMU
#define GT_WAVE_SYNC() emu::wave_rendezvous()
#else
#define GT_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

#define GT_LN_EPS 1e-5f

// A 16-byte zero page.  Out-of-range staging loads select THIS ADDRESS instead of zero-selecting the
// loaded value: a select after the load makes the compiler wait (vmcnt) right behind the load and
// kills the prefetch/compute overlap; a select before it costs one v_cndmask on the address.
// Deliberately non-const so the loads cannot be folded back into a value select.
__device__ __attribute__((aligned(16))) static float gt_zero_page[4] = {0.f, 0.f, 0.f, 0.f};
// hipcc addresses this symbol through the GOT (s_getpc + s_load_dwordx2 + s_waitcnt lgkmcnt(0)) and, used directly inside a
// staging loop, re-issues that scalar-memory round trip per use.  Kernels therefore fetch the address ONCE (gt_zero_ptr())
// and hand the pointer (`zp`) to the helpers.  (Making the value opaque with an empty asm was measured 10 % SLOWER on the
// whole step -- 466 vs 420 us -- so it is a plain read that the compiler is free to hoist.)
__device__ __forceinline__ const float* gt_zero_ptr() { return gt_zero_page; }

// ---- optional per-launch timing (gt_profile_*), by kernel class ----
// Off by default (zero overhead beyond one branch).  bench.py switches it on for a separate eager pass to measure the
// dominant kernel's average duration live.  The launch goes through hipExtLaunchKernelGGL with a start and a stop event:
// those take the dispatch packet's OWN begin/end timestamps -- the clock rocprofv3's kernel trace reads -- whereas
// hipEventRecord pairs around a launch add ~2.2 us of dispatch latency (38 % on a 5.8 us kernel).
struct GtProfile {
  bool on = false;
  const char* label = "";      // set by the caller just before a launch
  double flops = 0, bytes = 0; // algorithmic work of the next launch
};
extern GtProfile g_prof;
void gt_prof_events(hipEvent_t* start, hipEvent_t* stop);   // new record for the launch that follows
static inline void gt_prof_tag(const char* label, double flops, double bytes) {
  if (g_prof.on) { g_prof.label = label; g_prof.flops = flops; g_prof.bytes = bytes; }
}

template <typename... KArgs, typename... Args>
static inline void gt_launch(void (*kern)(KArgs...), dim3 grid, dim3 block, hipStream_t s, Args... args) {
#ifdef GT_EMU
  (void)s;
  emu::launch(grid, block, 0, [=]() { kern(args...); });
#else
  if (g_prof.on) {
    hipEvent_t a, b;
    gt_prof_events(&a, &b);
    hipExtLaunchKernelGGL<KArgs...>(kern, grid, block, 0u, s, a, b, 0u, static_cast<KArgs>(args)...);
  } else {
    kern<<<grid, block, 0, s>>>(args...);
  }
#endif
}

// ---- dropout RNG (spec in include/groove_hip.h) ------------------------------------------------
__host__ __device__ static inline uint32_t gt_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
struct DropArgs {
  const gt_step_state* st;   // nullptr -> disabled
  uint32_t site;
  uint32_t thr;              // p * 2^24; 0 -> disabled
  float scale;               // 1/(1-p)
};
__device__ static inline uint32_t gt_drop_key(const DropArgs& d) {
  if (d.thr == 0u || d.st == nullptr) return 0u;
  uint32_t s = d.st->seed_lo ^ gt_fmix32(d.st->step);
  uint32_t k = s ^ (d.site * 0x9E3779B9u);
  k = gt_fmix32(k) ^ d.st->seed_hi;
  return gt_fmix32(k + 0x7F4A7C15u);
}
// multiplier for element idx: scale if kept, 0 if dropped
__device__ static inline float gt_drop_mul(const DropArgs& d, uint32_t key, uint32_t idx) {
  if (d.thr == 0u || d.st == nullptr) return 1.0f;
#if defined(GT_ACCT_NOHASH)
  return (idx ^ key) == 0x12345u ? 0.0f : d.scale;      // (measurement build, WRONG RESULTS: what the per-element hash costs the one-kernel-per-op path)
#endif
  return ((gt_fmix32((idx * 0x9E3779B1u) ^ key) >> 8) >= d.thr) ? d.scale : 0.0f;
}

__device__ static inline float gt_wave_sum(float v) {
  v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);  v += __shfl_xor(v, 2);  v += __shfl_xor(v, 1);
  return v;
}
__device__ static inline float gt_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Partial results of many workgroups meeting in the LAST ARRIVER of a launch, without fences: the partials are write-through
// (agent-scope) stores, drained before one lane takes a ticket (agent-scope returning add); the workgroup whose add came last reads
// them with agent-scope loads, behind a workgroup barrier its ticket lane joins.  (MI355X_MICROARCH.md, valid hand-off forms, row 1.
// A __threadfence() pair instead writes back every dirty line of the XCD's L2 -- the activations just saved -- in every workgroup:
// 3.8 us of the headline's last forward phase.)
__device__ __forceinline__ void gt_pub_store(float* p, const float v) {
#ifdef GT_EMU
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ float gt_pub_load(const float* p) {
#ifdef GT_EMU
  return *p;
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ unsigned gt_pub_ticket(unsigned* counter) {       // (after this lane's seq_pub_store calls)
#ifdef GT_EMU
  return atomicAdd(counter, 1u);
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// scheduling hints (no-ops in the host emulator): GT_SGB = sched_group_barrier, GT_SCHED_FENCE = sched_barrier(0)
#ifdef GT_EMU
#define GT_SGB(mask, n)
#define GT_SCHED_FENCE()
#else
#define GT_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define GT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0);
#endif

// one (row, voice) element: the three loss terms, the hit-accuracy indicator and d loss / d (h, v, o)
template <bool WRT_LOGITS>
__device__ __forceinline__ void gt_loss_elem(const float h, const float v, const float o, const float yh, const float yv, const float yo,
                                             const float penalty, const float invM, float& bce, float& mv, float& mo, float& ok, float& dh,
                                             float& dv, float& dO) {
  const float pen = (yh == 1.0f) ? 1.0f : penalty;
  bce = (fmaxf(h, 0.f) - h * yh + log1pf(expf(-fabsf(h)))) * pen;
  mv = (v - yv) * (v - yv) * pen;
  mo = (o - yo) * (o - yo) * pen;
  ok = (((h > 0.f) ? 1.0f : 0.0f) == yh) ? 1.0f : 0.0f;      // sigmoid(h) > 0.5  <=>  h > 0
  dv = 2.0f * (v - yv) * pen * invM; dO = 2.0f * (o - yo) * pen * invM;
  if (WRT_LOGITS) { dv *= v * (1.0f - v); dO *= (0.5f - 2.0f * o * o); }
  dh = (gt_sigmoid(h) - yh) * pen * invM;
}

// The counters after an update: the dropout stream always moves on (the batch was consumed); Adam's t only when the update was APPLIED --
// with the exchange region's error word set (err[0]; nullptr: the caller has none) or a non-zero data-parallel guard element the update kernels
// apply nothing, and err[1] counts the skipped update instead (read by the host beside the word: StepEngine.skipped_updates).
__device__ __forceinline__ void gt_bump_counters(gt_step_state* st, unsigned* err, const float* guard) {
  const bool skip = (err != nullptr && err[0] != 0u) || (guard != nullptr && *guard != 0.f);
  st->step += 1u;
  if (!skip) st->opt_step += 1u;
  else if (err != nullptr) err[1] += 1u;
}
