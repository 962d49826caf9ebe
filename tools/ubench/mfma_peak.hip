// Micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate with no memory traffic at all -- what fraction of the 157.3 TF
// paper peak (256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz) the chip actually holds under a matrix-only load.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && tools/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, float* out, long long* clk) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

template <int NACC>
void run(int nwg, int iters, float* out, long long* clk, const char* tag) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) k<NACC><<<nwg, 256>>>(iters, out, clk);
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) k<NACC><<<nwg, 256>>>(iters, out, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double us = 1e3 * ms / reps;
  const double flops = (double)nwg * 4 /*waves*/ * iters * 4.0 * NACC * 2048.0;
  // wall_clock64 ticks at 100 MHz: shader clock = cycles / wall time
  const double mhz = h[1] > 0 ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;
  printf("%-34s wgs %5d: %9.1f us  %7.1f TFLOP/s  (%.1f %% of 157.3)  shader clock ~%.0f MHz  cycles/MFMA/SIMD %.1f\n", tag, nwg, us,
         flops / us / 1e6, 100.0 * flops / us / 1e6 / 157.3, mhz, (double)h[0] / ((double)iters * 4 * NACC));
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 64); hipMalloc(&clk, 64);
  run<16>(256, 4000, out, clk, "1 wave/SIMD, 16 accumulators");
  run<16>(512, 4000, out, clk, "2 waves/SIMD, 16 accumulators");
  run<16>(1024, 2000, out, clk, "4 waves/SIMD, 16 accumulators");
  run<4>(256, 16000, out, clk, "1 wave/SIMD, 4 accumulators");
  run<1>(256, 32000, out, clk, "1 wave/SIMD, 1 accumulator (dependent)");
  run<16>(256, 40000, out, clk, "1 wave/SIMD, 16 acc, 10x longer");
  run<16>(2560, 4000, out, clk, "10 rounds of 256 workgroups");
  return 0;
}
