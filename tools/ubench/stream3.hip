// Micro-benchmark 3: all workgroups stream the SAME 768 KB block; does the slab's memory shape matter?
//   contiguous: slab = 16 KB contiguous;  strided: slab = 128 segments of 128 B at a given row stride (a [N][K] weight slab)
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k(const float4* __restrict__ w, int nslab, int seg_f4, int stride_f4, float* out) {
  __shared__ float4 lds[1024];
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int s = 0; s < nslab; ++s) {
    float4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = i * 256 + tid;                      // 1024 chunks of 16 B = 16 KB per slab
      const int row = ch / seg_f4, c = ch % seg_f4;      // seg_f4 chunks per row segment
      r[i] = w[(size_t)row * stride_f4 + (size_t)s * seg_f4 + c];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) lds[i * 256 + tid] = r[i];
    __syncthreads();
    acc += lds[(tid * 7) & 1023].x;
    __syncthreads();
  }
  if (acc == 123.456f) out[0] = acc;
}

void run(const float4* w, float* out, int nwg, int seg_bytes, int stride_bytes, const char* tag) {
  const int total = 768 * 1024, nslab = total / 16384;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) k<<<nwg, 256>>>(w, nslab, seg_bytes / 16, stride_bytes / 16, out);
  hipEventRecord(a);
  for (int it = 0; it < 20; ++it) k<<<nwg, 256>>>(w, nslab, seg_bytes / 16, stride_bytes / 16, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / 20;
  printf("%-44s wgs %4d: %7.2f us -> %6.1f GB/s per WG\n", tag, nwg, us, total / us / 1e3);
}

int main() {
  float4* w; float* out;
  hipMalloc(&w, 64 << 20); hipMalloc(&out, 64);
  hipMemset(w, 0, 64 << 20);
  for (int nwg : {1, 128, 256}) {
    run(w, out, nwg, 16384, 16384, "contiguous 16 KB slabs");
    run(w, out, nwg, 128, 512, "128 rows x 128 B, row stride 512 B  (K=128)");
    run(w, out, nwg, 128, 2048, "128 rows x 128 B, row stride 2 KB   (K=512)");
    run(w, out, nwg, 128, 2048 + 128, "128 rows x 128 B, row stride 2 KB+128 B");
    run(w, out, nwg, 512, 2048, "32 rows x 512 B, row stride 2 KB (NN, F=512)");
    run(w, out, nwg, 512, 512, "32 rows x 512 B, row stride 512 B (NN, d=128)");
  }
  return 0;
}
