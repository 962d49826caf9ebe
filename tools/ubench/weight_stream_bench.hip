// What do 128 workgroups (one per CU: each asks for 140 KB of LDS, like the sequence-resident phase kernels) get out of L2 when ALL of
// them stream the SAME weight set -- one encoder layer of the headline model, 786 KB in fragment order -- at the same time, and does it
// change with 16 waves per workgroup instead of 8, or with more loads in flight per wave?  Wall time by HIP events over REP passes.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/weight_stream_bench.hip -o tools/ubench/weight_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int LAYER_BYTES = (4 * 128 * 128 + 2 * 128 * 512) * 4;       // in-proj + out-proj + FFN1 + FFN2 of d_model 128 / F 512

// every wave-instruction fetches 1 KB contiguous (a fragment of the packs); a wave keeps DEPTH groups of 8 instructions (one 16-column
// tile at K = 128) in flight; MFMAS > 0 puts that many v_mfma_f32_16x16x4_f32 behind each group (the kernels have 32 per tile)
template <int NT, int DEPTH, int MFMAS, bool ACC2 = false>
__global__ __launch_bounds__(NT) void stream(const float* W, float* sink, int rep) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = NT / 64;
  const int ngroups = LAYER_BYTES / 8192;                                 // 8 KB per group: 96 groups per pass
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 macc = {0.f, 0.f, 0.f, 0.f}, macc2 = {0.f, 0.f, 0.f, 0.f};
  f4 v[DEPTH][8];
  for (int r = 0; r < rep; ++r) {
    int g = wave;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int gg = g + d * nw < ngroups ? g + d * nw : g;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[d][i] = *reinterpret_cast<const f4*>(W + (size_t)gg * 2048 + i * 256 + lane * 4);
    }
    for (; g < ngroups; g += nw * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        f4 cur[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = v[d][i];
        const int gn = g + (d + DEPTH) * nw;
        const int gg = gn < ngroups ? gn : g;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[d][i] = *reinterpret_cast<const f4*>(W + (size_t)gg * 2048 + i * 256 + lane * 4);
        if (MFMAS > 0) {
#pragma unroll
          for (int m = 0; m < MFMAS; ++m) {
            if (ACC2 && (m & 1)) macc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[m & 7][m & 3], cur[(m + 1) & 7][m & 3], macc2, 0, 0, 0);
            else macc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[m & 7][m & 3], cur[(m + 1) & 7][m & 3], macc, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc += cur[i];
        }
      }
    }
  }
  acc += macc + macc2;
  if (acc.x == 1.2345e-30f) sink[tid] = acc.y + lds[tid];
}

template <int NT, int DEPTH, int MFMAS, bool ACC2 = false>
static void run(const float* W, float* sink, int nwg, const char* what) {
  const int rep = 50;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<NT, DEPTH, MFMAS, ACC2>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((stream<NT, DEPTH, MFMAS, ACC2>), dim3(nwg), dim3(NT), 140 * 1024, 0, W, sink, 5);
  hipEventRecord(a);
  hipLaunchKernelGGL((stream<NT, DEPTH, MFMAS, ACC2>), dim3(nwg), dim3(NT), 140 * 1024, 0, W, sink, rep);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us_pass = 1e3 * ms / rep, bpc = (double)LAYER_BYTES / (us_pass * 1e-6 * 2.4e9);
  printf("  %3d workgroups x %2d waves, %d tiles in flight per wave, %2d MFMAs per tile: %7.2f us per 786 KB pass  %5.1f B/clk/CU (2.4 GHz)  %5.2f TB/s chip-wide  %s\n",
         nwg, NT / 64, DEPTH, MFMAS, us_pass, bpc, nwg * (double)LAYER_BYTES / us_pass * 1e-6, what);
}

int main() {
  float* W; float* sink;
  hipMalloc(&W, 4u << 20); hipMalloc(&sink, 1 << 20);
  hipMemset(W, 0, 4u << 20);
  for (int nwg : {128}) {
    run<512, 1, 0>(W, sink, nwg, "");
    run<512, 2, 0>(W, sink, nwg, "(the kernels' depth)");
    run<512, 4, 0>(W, sink, nwg, "");
    run<1024, 1, 0>(W, sink, nwg, "");
    run<1024, 2, 0>(W, sink, nwg, "");
    run<512, 2, 32>(W, sink, nwg, "(+ the MFMAs of a 16-row tile)");
    run<1024, 2, 32>(W, sink, nwg, "");
    run<512, 2, 64>(W, sink, nwg, "(+ the MFMAs of a 32-row tile)");
    run<512, 1, 32>(W, sink, nwg, "");
    run<512, 3, 32>(W, sink, nwg, "");
    run<512, 4, 32>(W, sink, nwg, "");
    run<512, 2, 32, true>(W, sink, nwg, "(two accumulator chains)");
    run<256, 2, 32>(W, sink, nwg, "(4 waves)");
    run<256, 4, 32>(W, sink, nwg, "(4 waves)");
    run<256, 4, 32, true>(W, sink, nwg, "(4 waves, two accumulator chains)");
  }
  return 0;
}
