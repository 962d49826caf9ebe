// TEST INFRASTRUCTURE ONLY -- a single-threaded fiber emulator of the HIP execution model.
//
// Purpose: compile transformergrooveinfilling_amd/csrc/groove_hip.hip as plain host C++
// (-DGT_EMU) so that the kernels' index math, LDS staging, MFMA fragment maps and barrier
// structure can be debugged and sanitised (ASan/UBSan) in the build container, which has no GPU.
// It is built into tests/emu/libgroove_emu.so by tests/emu/build_emu.sh and loaded ONLY by
// tests that pass its path explicitly.  The product package never loads it and has no CPU
// fallback: transformergrooveinfilling_amd/_lib.py fails loudly when libgroove_hip.so is absent.
//
// Model: every thread of a workgroup is a ucontext fiber; fibers run round-robin and yield at
// __syncthreads() and at wave-level collectives (shuffles, MFMA).  v_mfma_f32_16x16x4_f32 is
// emulated as the k-ordered fmaf chain the hardware performs (cdna_hip_programming.md 3,
// "FP32-input MFMA"), with the documented lane maps A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[row=4*(l>>4)+reg][col=l&15].
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct uint2 { uint32_t x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
typedef float f32x4 __attribute__((vector_size(16)));
typedef int hipStream_t_dummy;
typedef void* hipStream_t;
typedef int hipError_t;
typedef void* hipEvent_t;
#define hipSuccess 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

namespace emu {
extern dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
extern unsigned char* dyn_smem_;
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void block_sync();
// A workgroup that waits for data ANOTHER workgroup of the same launch publishes (the pair exchange of gt_seq.h's QUAD forward): the
// hardware runs both at once and the waiter spins; the emulator runs workgroups one after the other, so the waiter gives up here --
// every thread of the block calls this at the same program point, none returns -- and launch() runs the block again from the start
// after the others (its writes so far are repeated with the same values).  launch_serial: a per-launch number for the exchange tags.
[[noreturn]] void block_retry();
extern unsigned launch_serial;
// ds_read_b64_tr_b16 (gfx950): per group of 16 consecutive lanes a 4-row x 16-column block of 16-bit elements, delivered column-major --
// lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3; lane i receives column i, row q in its element q
struct tr16x4 { uint16_t v[4]; };
tr16x4 lds_tr16(const uint16_t* addr);
void wave_rendezvous();     // all live lanes of the calling wave (hardware: lanes run in lock-step; the emulator: fibers do not)
float wave_shfl(float v, int src_lane);
f32x4 mfma16(float a, float b, f32x4 c);
}  // namespace emu

#define threadIdx emu::threadIdx_
#define blockIdx emu::blockIdx_
#define blockDim emu::blockDim_
#define gridDim emu::gridDim_

static inline void __syncthreads() { emu::block_sync(); }
static inline float __shfl_xor(float v, int m) {
  int lane = (threadIdx.x + threadIdx.y * blockDim.x) & 63;
  return emu::wave_shfl(v, lane ^ m);
}
static inline float __shfl(float v, int src) { return emu::wave_shfl(v, src & 63); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline void __threadfence() {}
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
#define GT_MFMA16(a, b, c) emu::mfma16((a), (b), (c))
typedef float f32x16 __attribute__((vector_size(64)));
namespace emu { f32x16 mfma32(float a, float b, f32x16 c); }
#define GT_MFMA32(a, b, c) emu::mfma32((a), (b), (c))
// bf16 operands (gt_config.precision = 1): 8 bf16 per lane, fp32 accumulate
struct bf16x8 { uint16_t v[8]; };
namespace emu { f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c); }
#define GT_MFMA16_BF16(a, b, c) emu::mfma16_bf16((a), (b), (c))
namespace emu { f32x16 mfma32_bf16(bf16x8 a, bf16x8 b, f32x16 c); }
#define GT_MFMA32_BF16(a, b, c) emu::mfma32_bf16((a), (b), (c))
static inline uint16_t gt_f2bf(float f) {          // round to nearest even; NaN stays NaN (what v_cvt_pk_bf16_f32 does)
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

#define GT_BF16X8_SET(vec, j, x) (vec).v[j] = gt_f2bf(x)
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

#ifdef GT_EMU_IMPL
namespace emu {
dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
unsigned char* dyn_smem_ = nullptr;

namespace {
enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber { ucontext_t ctx; char* stack; int state; long nbar; };
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
int cur = -1, nthreads = 0;
const std::function<void()>* body_ = nullptr;
float scratchA[16][64], scratchB[16][64];

void yield_to_sched() { swapcontext(&fibers[cur].ctx, &sched_ctx); }
void trampoline() {
  (*body_)();
  fibers[cur].state = DONE;
  swapcontext(&fibers[cur].ctx, &sched_ctx);
}
void set_tid(int t) {
  threadIdx_.x = t % blockDim_.x;
  threadIdx_.y = (t / blockDim_.x) % blockDim_.y;
  threadIdx_.z = t / (blockDim_.x * blockDim_.y);
}
void run_block() {
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = fibers[t];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &sched_ctx;
    f.state = RUN;
    f.nbar = 0;
    makecontext(&f.ctx, trampoline, 0);
  }
  int nwaves = (nthreads + 63) / 64;
  for (;;) {
    int done = 0, progressed = 0;
    for (int t = 0; t < nthreads; ++t) {
      if (fibers[t].state == DONE) { ++done; continue; }
      if (fibers[t].state != RUN) continue;
      cur = t;
      set_tid(t);
      swapcontext(&sched_ctx, &fibers[t].ctx);
      ++progressed;
    }
    if (done == nthreads) {
      // hardware counts exited waves out of a barrier, so a mismatch would not hang -- it would silently mis-synchronise
      for (int t = 1; t < nthreads; ++t)
        if (fibers[t].nbar != fibers[0].nbar) {
          fprintf(stderr, "hip_emu: threads of one workgroup executed different numbers of barriers (%ld vs %ld, thread %d)\n",
                  fibers[0].nbar, fibers[t].nbar, t);
          abort();
        }
      break;
    }
    // release barriers whose participants have all arrived
    int wb = 0, live = 0;
    for (int t = 0; t < nthreads; ++t) {
      if (fibers[t].state != DONE) ++live;
      if (fibers[t].state == WAIT_BLOCK) ++wb;
    }
    bool released = false;
    if (live > 0 && wb == live) {
      for (int t = 0; t < nthreads; ++t) if (fibers[t].state == WAIT_BLOCK) fibers[t].state = RUN;
      released = true;
    }
    for (int w = 0; w < nwaves; ++w) {
      int ww = 0, wl = 0;
      for (int t = w * 64; t < nthreads && t < w * 64 + 64; ++t) {
        if (fibers[t].state != DONE) ++wl;
        if (fibers[t].state == WAIT_WAVE) ++ww;
      }
      if (wl > 0 && ww == wl) {
        for (int t = w * 64; t < nthreads && t < w * 64 + 64; ++t) if (fibers[t].state == WAIT_WAVE) fibers[t].state = RUN;
        released = true;
      }
    }
    if (!released && !progressed) {
      fprintf(stderr, "hip_emu: deadlock (divergent barrier?) block=(%u,%u,%u)\n", blockIdx_.x, blockIdx_.y, blockIdx_.z);
      abort();
    }
  }
}
void wave_sync() { fibers[cur].state = WAIT_WAVE; yield_to_sched(); }
}  // namespace

void wave_rendezvous() { wave_sync(); }
static bool retry_ = false;
unsigned launch_serial = 0;
void block_retry() {
  retry_ = true;
  fibers[cur].state = DONE;
  swapcontext(&fibers[cur].ctx, &sched_ctx);
  abort();                                        // (never resumed)
}
void block_sync() { fibers[cur].state = WAIT_BLOCK; ++fibers[cur].nbar; yield_to_sched(); }

float wave_shfl(float v, int src_lane) {
  int w = cur / 64, l = cur % 64;
  scratchA[w][l] = v;
  wave_sync();
  float r = scratchA[w][src_lane];
  wave_sync();
  return r;
}

static const uint16_t* scratchP[16][64];
tr16x4 lds_tr16(const uint16_t* addr) {
  int w = cur / 64, l = cur % 64;
  scratchP[w][l] = addr;
  wave_sync();
  const int g0 = l & ~15, i = l & 15;
  tr16x4 r;
  for (int q = 0; q < 4; ++q) r.v[q] = scratchP[w][g0 + 4 * q + (i >> 2)][i & 3];
  wave_sync();
  return r;
}

f32x4 mfma16(float a, float b, f32x4 c) {
  int w = cur / 64, l = cur % 64;
  scratchA[w][l] = a;
  scratchB[w][l] = b;
  wave_sync();
  int col = l & 15, g = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * g + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(scratchA[w][row + 16 * k], scratchB[w][col + 16 * k], acc);
    c[r] = acc;
  }
  wave_sync();
  return c;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D[row=(reg&3)+8*(reg>>2)+4*(l>>5)][col=l&31]; k-ordered fmaf chain
f32x16 mfma32(float a, float b, f32x16 c) {
  int w = cur / 64, l = cur % 64;
  scratchA[w][l] = a;
  scratchB[w][l] = b;
  wave_sync();
  int col = l & 31, hh = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(scratchA[w][row + 32 * k], scratchB[w][col + 32 * k], acc);
    c[r] = acc;
  }
  wave_sync();
  return c;
}

// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][col l&15]; products of two bf16 are exact in fp32, the
// 32-term sum is taken in double and rounded once (the hardware's internal accumulation order is not specified: tests
// compare within a tolerance)
static uint16_t scratchHA[16][64][8], scratchHB[16][64][8];
f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  int w = cur / 64, l = cur % 64;
  for (int j = 0; j < 8; ++j) { scratchHA[w][l][j] = a.v[j]; scratchHB[w][l][j] = b.v[j]; }
  wave_sync();
  int col = l & 15, g = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * g + r;
    double acc = c[r];
    for (int kg = 0; kg < 4; ++kg)
      for (int j = 0; j < 8; ++j) {
        uint32_t ua = (uint32_t)scratchHA[w][row + 16 * kg][j] << 16, ub = (uint32_t)scratchHB[w][col + 16 * kg][j] << 16;
        float fa, fb;
        memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
        acc += (double)fa * (double)fb;
      }
    c[r] = (float)acc;
  }
  wave_sync();
  return c;
}

// v_mfma_f32_32x32x16_bf16: A[row l&31][k = 8(l>>5) + j], B[k = 8(l>>5) + j][col l&31]; D as the f32 32x32 form
f32x16 mfma32_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  int w = cur / 64, l = cur % 64;
  for (int j = 0; j < 8; ++j) { scratchHA[w][l][j] = a.v[j]; scratchHB[w][l][j] = b.v[j]; }
  wave_sync();
  int col = l & 31, hh = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
    double acc = c[r];
    for (int kh = 0; kh < 2; ++kh)
      for (int j = 0; j < 8; ++j) {
        uint32_t ua = (uint32_t)scratchHA[w][row + 32 * kh][j] << 16, ub = (uint32_t)scratchHB[w][col + 32 * kh][j] << 16;
        float fa, fb;
        memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
        acc += (double)fa * (double)fb;
      }
    c[r] = (float)acc;
  }
  wave_sync();
  return c;
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  nthreads = block.x * block.y * block.z;
  if (nthreads > 1024 || (nthreads % 64) != 0) { fprintf(stderr, "hip_emu: bad block size %d\n", nthreads); abort(); }
  if (smem > 160 * 1024) { fprintf(stderr, "hip_emu: dynamic LDS %zu > 160 KiB\n", smem); abort(); }
  if ((int)fibers.size() < nthreads) {
    size_t old = fibers.size();
    fibers.resize(nthreads);
    for (size_t t = old; t < fibers.size(); ++t) fibers[t].stack = (char*)malloc(kStack);
  }
  std::vector<unsigned char> dyn(smem + 16, 0xA5);   // poison: uninitialised LDS reads show up
  dyn_smem_ = dyn.data();
  blockDim_ = block;
  gridDim_ = grid;
  body_ = &body;
  ++launch_serial;
  std::vector<dim3> pending, again;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) pending.push_back(dim3(x, y, z));
  int stalled = 0;                                   // passes in a row that completed no workgroup
  while (!pending.empty()) {
    again.clear();
    for (const dim3& blk : pending) {
      blockIdx_ = blk;
      retry_ = false;
      run_block();
      if (retry_) again.push_back(blk);           // waited for a workgroup that has not run yet: once more after the others
    }
    // (a launch with k meetings per workgroup pair can pass k - 1 times without a completion: each pass takes every pair one meeting further)
    if (again.size() == pending.size()) {
      if (++stalled > 4) { fprintf(stderr, "hip_emu: every remaining workgroup waits for another one (%zu blocks)\n", again.size()); abort(); }
    } else stalled = 0;
    pending.swap(again);
  }
  dyn_smem_ = nullptr;
}
}  // namespace emu
#endif  // GT_EMU_IMPL
