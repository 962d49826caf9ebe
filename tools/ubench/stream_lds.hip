// Micro-benchmark: how fast can ONE workgroup per CU stream an L2-resident weight block (shared by all workgroups)
// global -> registers -> LDS, as a function of the prefetch depth (slabs in flight) and slab size?
// usage: ./stream_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int DEPTH, int PER>   // PER float4 per thread per slab (slab = 256*PER*16 B)
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ w, int nslab, float* out) {
  __shared__ float4 lds[2][256 * PER];
  const int tid = threadIdx.x;
  float4 r[DEPTH][PER];
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < PER; ++i) r[d][i] = w[(size_t)d * 256 * PER + i * 256 + tid];
  for (int s = 0; s < nslab; s += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int i = 0; i < PER; ++i) lds[(s + d) & 1][i * 256 + tid] = r[d][i];
      const int nx = s + d + DEPTH;
      if (nx < nslab) {
#pragma unroll
        for (int i = 0; i < PER; ++i) r[d][i] = w[(size_t)nx * 256 * PER + i * 256 + tid];
      }
      __syncthreads();
      acc += lds[(s + d) & 1][(tid * 7) & (256 * PER - 1)].x;     // consume something
      __syncthreads();
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int DEPTH, int PER>
void run(const float4* w, float* out, int total_f4, int nwg) {
  const int nslab = total_f4 / (256 * PER);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) stream_kernel<DEPTH, PER><<<nwg, 256>>>(w, nslab, out);
  hipEventRecord(a);
  const int reps = 20;
  for (int it = 0; it < reps; ++it) stream_kernel<DEPTH, PER><<<nwg, 256>>>(w, nslab, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, bytes = (double)total_f4 * 16;
  printf("wgs %4d depth %d slab %3d KB: %7.2f us/launch  -> %6.1f GB/s per WG, %6.2f TB/s aggregate\n", nwg, DEPTH, PER * 4, us,
         bytes / us / 1e3, bytes * nwg / us / 1e6);
}

int main() {
  const int total_f4 = 768 * 1024 / 16;          // 768 KB weight block
  float4* w; float* out;
  hipMalloc(&w, total_f4 * 16); hipMalloc(&out, 64);
  hipMemset(w, 0, total_f4 * 16);
  for (int nwg : {128, 256, 512}) {
    run<1, 8>(w, out, total_f4, nwg);
    run<2, 8>(w, out, total_f4, nwg);
    run<4, 8>(w, out, total_f4, nwg);
    run<1, 4>(w, out, total_f4, nwg);
    run<2, 4>(w, out, total_f4, nwg);
    run<4, 4>(w, out, total_f4, nwg);
    run<8, 2>(w, out, total_f4, nwg);
  }
  return 0;
}
