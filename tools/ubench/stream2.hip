// Micro-benchmark 2: isolate what bounds per-CU streaming: shared vs private source, LDS write or not, threads per WG.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NT, int PER, bool LDSW>
__global__ __launch_bounds__(NT) void k(const float4* __restrict__ w, int nslab, size_t wg_stride, float* out) {
  __shared__ float4 lds[NT * PER];
  const int tid = threadIdx.x;
  const float4* src = w + blockIdx.x * wg_stride;
  float acc = 0.f;
  for (int s = 0; s < nslab; ++s) {
    float4 r[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) r[i] = src[(size_t)s * NT * PER + i * NT + tid];
    if (LDSW) {
#pragma unroll
      for (int i = 0; i < PER; ++i) lds[i * NT + tid] = r[i];
      __syncthreads();
      acc += lds[(tid * 7) & (NT * PER - 1)].x;
      __syncthreads();
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) acc += r[i].x + r[i].w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int NT, int PER, bool LDSW>
void run(const float4* w, float* out, int total_f4, int nwg, bool priv, const char* tag) {
  const int nslab = total_f4 / (NT * PER);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const size_t stride = priv ? total_f4 : 0;
  for (int it = 0; it < 3; ++it) k<NT, PER, LDSW><<<nwg, NT>>>(w, nslab, stride, out);
  hipEventRecord(a);
  const int reps = 20;
  for (int it = 0; it < reps; ++it) k<NT, PER, LDSW><<<nwg, NT>>>(w, nslab, stride, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, bytes = (double)total_f4 * 16;
  printf("%-28s wgs %4d thr %4d per %d: %7.2f us -> %6.1f GB/s per WG, %6.2f TB/s aggregate\n", tag, nwg, NT, PER, us, bytes / us / 1e3,
         bytes * nwg / us / 1e6);
}

int main() {
  const int total_f4 = 768 * 1024 / 16;
  float4* w; float* out;
  hipMalloc(&w, (size_t)total_f4 * 16 * 256); hipMalloc(&out, 64);
  hipMemset(w, 0, (size_t)total_f4 * 16 * 256);
  for (int nwg : {128, 256}) {
    run<256, 8, true>(w, out, total_f4, nwg, false, "shared src, LDS write");
    run<256, 8, false>(w, out, total_f4, nwg, false, "shared src, regs only");
    run<256, 8, true>(w, out, total_f4, nwg, true, "private src, LDS write");
    run<256, 8, false>(w, out, total_f4, nwg, true, "private src, regs only");
    run<512, 4, false>(w, out, total_f4, nwg, false, "shared src, regs, 512 thr");
    run<1024, 2, false>(w, out, total_f4, nwg, false, "shared src, regs, 1024 thr");
    run<1024, 4, false>(w, out, total_f4, nwg, false, "shared src, regs, 1024x4");
    run<256, 16, false>(w, out, total_f4, nwg, false, "shared src, regs, per 16");
  }
  run<256, 8, false>(w, out, total_f4, 1, false, "ONE workgroup");
  run<256, 8, false>(w, out, total_f4, 8, false, "8 workgroups");
  run<256, 8, false>(w, out, total_f4, 32, false, "32 workgroups");
  return 0;
}
