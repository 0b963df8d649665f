// Does a wave64 VALU instruction cost less when one 32-lane half of the wave is switched off?
// (If the SIMD skipped an all-inactive half, packing a step's active lanes into one half would pay.)
//   hipcc --offload-arch=gfx950 -O3 -o ubench_halfwave tools/ubench_halfwave.hip && ./ubench_halfwave
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void chain(float* out, int firstLane, int lastLane, int iters) {
  const int lane = threadIdx.x & 63;
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  if (lane >= firstLane && lane <= lastLane) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_min_f32 %3, %3, %0\n v_fma_f32 %2, %2, %1, %0\n v_max_f32 %1, %1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, grid = cus * 5, iters = 4096;
  float* out;
  (void)hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int masks[][2] = {{0, 63}, {0, 31}, {32, 63}, {0, 15}, {0, 47}, {16, 47}, {0, 0}};
  for (auto& m : masks) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(chain, dim3(grid), dim3(256), 0, 0, out, m[0], m[1], iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double instr = (double)iters * 64;  // per wave
    printf("lanes %2d..%2d active: %8.3f ms  %.2f SIMD-cycles per wave-instruction at 2.4 GHz, 5 waves per SIMD\n", m[0], m[1], best,
           best * 1e-3 * 2.4e9 / (instr * 5));
  }
  return 0;
}
