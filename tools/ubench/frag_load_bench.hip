// How fast can ONE workgroup (8 waves, one CU) pull a weight matrix out of L2 into registers?  Two footprints per
// global_load_dwordx4 wave-instruction: "fragment" = what an MFMA B operand wants when loaded straight from a row-major
// [n][k] matrix (16 rows x 64 contiguous bytes), "line" = 1 KB contiguous (8 full 128-byte lines).  Prints bytes / cycle.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/frag_load_bench.hip -o gpurun_variants_frag_load_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memtime(); }

// K = 128 floats per row (512 B).  NL loads per wave, all issued back to back, then one wait.
template <int MODE, int NL>
__global__ __launch_bounds__(512) void k(const float* W, long long* out, float* sink, int nwaves) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= nwaves) return;
  const int l16 = lane & 15, lg = lane >> 4;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = now();
  f4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const float* p;
    if (MODE == 0) {          // fragment: tile t = wave + 8 * (i / 8), k-step u = i % 8: row 16 t + l16, bytes 64 u + 16 lg
      const int t = wave + 8 * (i / 8), u = i % 8;
      p = W + (size_t)(16 * t + l16) * 128 + 16 * u + 4 * lg;
    } else {                  // line: the same bytes of the same tiles, 1 KB contiguous per instruction
      const int t = wave + 8 * (i / 8), u = i % 8;
      p = W + (size_t)(16 * t) * 128 + u * 256 + lane * 4;
    }
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NL; ++i) { asm volatile("" : "+v"(v[i])); acc += v[i]; }
  const long long t1 = now();
  if (lane == 0) { out[2 * wave] = t0; out[2 * wave + 1] = t1; }
  if (acc.x == 1.2345e-30f) sink[tid] = acc.y;
}

template <int MODE, int NL>
void run(const float* W, long long* out, float* sink, int nwaves, int nwg, const char* name) {
  long long h[16];
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((k<MODE, NL>), dim3(nwg), dim3(512), 0, 0, W, out, sink, nwaves);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 16 * 8, hipMemcpyDeviceToHost);
    long long lo = h[0], hi = h[1];
    for (int w = 0; w < nwaves; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    if (rep > 0 && (double)(hi - lo) < best) best = (double)(hi - lo);
  }
  const double bytes = (double)nwaves * NL * 1024;
  printf("  %-10s %d waves x %2d loads, %3d workgroups: %7.0f cycles  %6.1f B/clk/CU  %5.0f cycles per wave-instruction\n", name, nwaves, NL, nwg, best,
         bytes / best, best / NL);
}

int main() {
  float* W; long long* out; float* sink;
  hipMalloc(&W, 64u << 20); hipMalloc(&out, 4096); hipMalloc(&sink, 1 << 20);
  hipMemset(W, 0, 64u << 20);
  for (int nwg : {1, 64}) {
    for (int nw : {1, 8}) {
      run<0, 8>(W, out, sink, nw, nwg, "fragment"); run<1, 8>(W, out, sink, nw, nwg, "line");
      run<0, 24>(W, out, sink, nw, nwg, "fragment"); run<1, 24>(W, out, sink, nw, nwg, "line");
    }
  }
  return 0;
}
