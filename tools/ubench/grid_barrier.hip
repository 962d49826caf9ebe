// Micro-benchmark: cost of a device-wide barrier inside ONE persistent kernel (sense-reversing counter, agent-scope
// atomics) versus the ~3.9 us a kernel boundary costs in a stream / hipGraph on this system.  One workgroup per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/grid_barrier.hip -o tools/ubench/grid_barrier && tools/ubench/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nwg, unsigned& my_gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                           // release this workgroup's writes
    const unsigned g = my_gen;
    if (atomicAdd(count, 1u) == nwg - 1) {                     // last arrival: reset and open the next generation
      *count = 0u;
      __threadfence();
      atomicExch((unsigned*)gen, g + 1u);
    } else {
      while (__hip_atomic_load((unsigned*)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();                                           // acquire the others' writes
  }
  my_gen += 1u;
  __syncthreads();
}

__global__ __launch_bounds__(256) void k(unsigned* count, unsigned* gen, int iters, float* data, float* out) {
  unsigned my_gen = 0u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // a token amount of dependent work per phase: every workgroup writes a slot, then reads its neighbour's after the barrier
    if (threadIdx.x == 0) data[blockIdx.x] = acc + (float)it;
    grid_barrier(count, gen, gridDim.x, my_gen);
    if (threadIdx.x == 0) acc += __hip_atomic_load(&data[(blockIdx.x + 1) % gridDim.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 1e-9f;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ void nullk(float* out) { if (out == nullptr) out[0] = 0.f; }

int main() {
  unsigned *count, *gen; float *data, *out;
  hipMalloc(&count, 4); hipMalloc(&gen, 4); hipMalloc(&data, 4096 * 4); hipMalloc(&out, 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int nwg : {64, 128, 256, 512}) {
    const int iters = 2000;
    hipMemset(count, 0, 4); hipMemset(gen, 0, 4);
    k<<<nwg, 256>>>(count, gen, 10, data, out);                // warm-up
    hipDeviceSynchronize();
    hipMemset(count, 0, 4); hipMemset(gen, 0, 4);
    hipEventRecord(a);
    k<<<nwg, 256>>>(count, gen, iters, data, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("grid barrier, %4d workgroups x 256 threads: %6.2f us per barrier\n", nwg, 1e3 * ms / iters);
  }
  hipEventRecord(a);
  for (int i = 0; i < 2000; ++i) nullk<<<1, 64>>>(out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("back-to-back empty kernels in one stream:      %6.2f us per launch\n", 1e3 * ms / 2000);
  return 0;
}
