// Micro-benchmark over the REAL kernels of gt_gemm.h: one GEMM shape, several tile configurations, TFLOP/s each.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I transformergrooveinfilling_amd/csrc tools/ubench/gemm_bench.hip -o tools/ubench/gemm_bench
//   tools/ubench/gemm_bench [M N K]        (default: the d_model 512 QKV projection at 16384 tokens)
#include "gt_gemm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

GtProfile g_prof;
void gt_prof_events(hipEvent_t*, hipEvent_t*) {}

template <int WM, int WN, int TM, int TN, int BK, bool AKM, bool BKM, int EPI>
static void run(const char* tag, GemmArgs g, int reps) {
  typedef GemmCfg<WM, WN, TM, TN, BK, AKM, BKM, EPI> Cfg;
  g.k_chunk = (g.K + BK - 1) / BK * BK;
  dim3 grid((g.N + Cfg::BN - 1) / Cfg::BN, (g.M + Cfg::BM - 1) / Cfg::BM, 1);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) gemm_kernel<WM, WN, TM, TN, BK, AKM, BKM, EPI><<<grid, dim3(Cfg::NT)>>>(g);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) gemm_kernel<WM, WN, TM, TN, BK, AKM, BKM, EPI><<<grid, dim3(Cfg::NT)>>>(g);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, tf = 2.0 * g.M * g.N * g.K / us / 1e6;
  printf("%-44s grid %5d x %3d  LDS %6d B  %8.1f us  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", tag, grid.y, grid.x, (int)(Cfg::SMEM * 4), us, tf,
         100.0 * tf / 157.3);
}

template <bool BKM, int EPI>
static void run32(const char* tag, GemmArgs g, int reps) {
  dim3 grid(g.N / 128, g.M / 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) gemm32_kernel<BKM, EPI><<<grid, 256>>>(g);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) gemm32_kernel<BKM, EPI><<<grid, 256>>>(g);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, tf = 2.0 * g.M * g.N * g.K / us / 1e6;
  printf("%-44s grid %5d x %3d  LDS %6d B  %8.1f us  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", tag, grid.y, grid.x, (int)(Gemm32Cfg::smem<false, BKM>() * 4), us, tf,
         100.0 * tf / 157.3);
}

template <bool BKM, int EPI, int PREC>
static void run64(const char* tag, GemmArgs g, int reps) {
  dim3 grid(g.N / 64, g.M / 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) gemm64_kernel<BKM, EPI, PREC><<<grid, 256>>>(g);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) gemm64_kernel<BKM, EPI, PREC><<<grid, 256>>>(g);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, tf = 2.0 * g.M * g.N * g.K / us / 1e6;
  printf("%-44s grid %5d x %3d  LDS %6d B  %8.1f us  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", tag, grid.y, grid.x, (int)(Gemm64Cfg::SMEM * 4), us, tf,
         100.0 * tf / 157.3);
}
template <int WHICH>
static void runh(const char* tag, GemmArgs g, int reps) {
  dim3 grid = WHICH == 64 ? dim3(g.N / 64, g.M / 64) : dim3(g.N / 128, g.M / 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < reps + 3; ++i) {
    if (i == 3) hipEventRecord(a);
    if (WHICH == 64) gemm64h_kernel<EPI_STORE><<<grid, 256>>>(g); else gemm32h_kernel<EPI_STORE><<<grid, 256>>>(g);
  }
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, tf = 2.0 * g.M * g.N * g.K / us / 1e6;
  printf("%-44s grid %5d x %3d  %8.1f us  %6.1f TFLOP/s\n", tag, grid.y, grid.x, us, tf);
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 512;
  float *A, *B, *C, *bias;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&bias, (size_t)N * 4);
  std::vector<float> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.0f - 0.5f;
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)(N < M ? N : M) * K * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, (size_t)N * 4);
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.bias = bias;
  printf("C[%d,%d] = A[%d,%d] * B[%d,%d]^T   (NT: both operands k-contiguous, EPI_STORE)\n", M, N, M, K, N, K);
  const int reps = 20;
#ifdef GT_BENCH_NOEPI
  g.mask_scale = 12345.f;
  printf("(epilogue skipped: main loop only)\n");
#endif
  // bring the chip to its loaded clock first: the first kernels of a process otherwise read 5-10 % low
  for (int i = 0; i < 200; ++i) gemm_kernel<2, 2, 4, 4, 32, false, false, EPI_STORE><<<dim3((N + 127) / 128, (M + 127) / 128), 256>>>(g);
  hipDeviceSynchronize();
  if (gemm64_ok(g, EPI_STORE)) {
    run64<false, EPI_STORE, 0>("gt_gemm64.h 64x64 32x32x2 ring, NT store", g, reps);
    run64<false, EPI_RELU_DROP, 0>("gt_gemm64.h 64x64 32x32x2 ring, NT relu", g, reps);
    run64<false, EPI_STORE, 1>("gt_gemm64.h 64x64 fp32 source -> bf16 MFMA", g, reps);
    uint16_t *A16, *B16;
    hipMalloc(&A16, (size_t)M * K * 2); hipMalloc(&B16, (size_t)N * K * 2);
    hipMemset(A16, 0, (size_t)M * K * 2); hipMemset(B16, 0, (size_t)N * K * 2);
    GemmArgs gh = g; gh.A16 = A16; gh.B16 = B16; gh.lda16 = gh.ldb16 = K; gh.bf16 = 1;
    runh<64>("gt_gemm64.h 64x64 bf16 sources", gh, reps);
    if (gemm32h_ok(gh, EPI_STORE)) runh<128>("gt_gemm32.h 128x128 bf16 sources", gh, reps);
    runh<64>("gt_gemm64.h 64x64 bf16 sources", gh, reps);
  }
  if (gemm32_ok(g, EPI_STORE, false)) {
    run32<false, EPI_STORE>("gt_gemm32.h 128x128 32x32x2 ring, NT store", g, reps);
    run32<false, EPI_RELU_DROP>("gt_gemm32.h 128x128 32x32x2 ring, NT relu", g, reps);
  }
  run<2, 2, 4, 4, 32, false, false, EPI_STORE>("128x128 <2,2,4,4,BK32> (16x16x4, one-deep)", g, reps);
  if (gemm32_ok(g, EPI_STORE, false)) run32<false, EPI_STORE>("gt_gemm32.h 128x128 32x32x2 ring, NT store", g, reps);
  run<2, 2, 1, 1, 64, false, false, EPI_STORE>("32x32   <2,2,1,1,BK64>", g, reps);
  run<2, 2, 2, 2, 32, false, false, EPI_STORE>("64x64   <2,2,2,2,BK32>", g, reps);
  run<2, 2, 2, 2, 64, false, false, EPI_STORE>("64x64   <2,2,2,2,BK64>", g, reps);
  run<2, 2, 4, 4, 16, false, false, EPI_STORE>("128x128 <2,2,4,4,BK16>", g, reps);
  run<2, 2, 4, 4, 32, false, false, EPI_STORE>("128x128 <2,2,4,4,BK32>", g, reps);
  run<2, 2, 4, 4, 64, false, false, EPI_STORE>("128x128 <2,2,4,4,BK64>", g, reps);
  run<2, 4, 4, 2, 32, false, false, EPI_STORE>("128x128 <2,4,4,2,BK32> 8 waves", g, reps);
  run<4, 2, 4, 4, 32, false, false, EPI_STORE>("256x128 <4,2,4,4,BK32> 8 waves", g, reps);
  run<2, 4, 4, 4, 32, false, false, EPI_STORE>("128x256 <2,4,4,4,BK32> 8 waves", g, reps);
  run<4, 2, 4, 4, 16, false, false, EPI_STORE>("256x128 <4,2,4,4,BK16> 8 waves", g, reps);
  // NN (dgrad): B row-contiguous
  g.ldb = N;   // B[k*ldb + n]: reuse the buffer as a (K x N) matrix (N*K floats)
  printf("NN (dgrad layout: B[k][n])\n");
  if (gemm32_ok(g, EPI_STORE, true)) run32<true, EPI_STORE>("gt_gemm32.h 128x128 32x32x2 ring, NN store", g, reps);
  if (gemm64_ok(g, EPI_STORE)) run64<true, EPI_STORE, 0>("gt_gemm64.h 64x64 32x32x2 ring, NN store", g, reps);
  run<2, 2, 1, 1, 64, false, true, EPI_STORE>("32x32   <2,2,1,1,BK64>", g, reps);
  run<2, 2, 4, 4, 32, false, true, EPI_STORE>("128x128 <2,2,4,4,BK32>", g, reps);
  run<4, 2, 4, 4, 32, false, true, EPI_STORE>("256x128 <4,2,4,4,BK32> 8 waves", g, reps);
  return 0;
}
