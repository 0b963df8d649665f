// What would a node visit cost if the whole BVH sat in LDS?  One workgroup of 1024 threads per CU (16 waves, 4 per
// SIMD) chases random 32-byte records through a 128 KB LDS table: two ds_read_b128 per hop (the record), plus the
// traversal stack's ds_read_b32 + ds_write_b32 in [slot][thread] layout, on `activeLanes` of 64 lanes.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lds tools/ubench_lds.hip && ./ubench_lds
// Output: cycles per hop per wave and per CU (the CU figure is what compares with tools/ubench_tcp.hip's
// "clk/hop/CU" for the same hop through the vector L1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool STACK>
__global__ __launch_bounds__(1024) void chase(uint32_t tableBytes, int activeLanes, int hops, uint32_t* out, unsigned long long* clocks) {
  extern __shared__ uint32_t lds[];
  const uint32_t records = tableBytes / 32;
  // first dword of every record: a pseudo-random next index; the rest zero
  for (uint32_t i = threadIdx.x; i < tableBytes / 4; i += blockDim.x) {
    uint32_t x = (i >> 3) * 0x9E3779B1u + blockIdx.x;
    x ^= x >> 15;
    x *= 0x2545F491u;
    lds[i] = (i & 7) == 0 ? (x >> 9) & (records - 1) : 0;
  }
  uint32_t* stack = lds + tableBytes / 4 + threadIdx.x;  // [slot][thread], 8 slots of uint16 pairs would halve this
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t acc = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (lane < activeLanes) {
    uint32_t rec = (threadIdx.x * 0x9E3779B1u >> 7) & (records - 1);
    int sp = 0;
    for (int k = 0; k < hops; ++k) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(lds + rec * 8), b = *reinterpret_cast<const u32x4*>(lds + rec * 8 + 4);
      if (STACK) {
        const uint32_t top = stack[sp * 1024];
        stack[((sp + 1) & 3) * 1024] = a.x;
        sp = (sp + (int)(a.x & 1)) & 3;
        acc += top & 1;
      }
      rec = (a.x + b.y + (acc & 0)) & (records - 1);
    }
    acc += rec;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = c1 - c0;
    clocks[1] = r1 - r0;
  }
}

template <bool STACK>
static void run(uint32_t tableBytes, int lanes, int threads, int cus, uint32_t* out, unsigned long long* clocks) {
  const int hops = 8192;
  const size_t ldsBytes = tableBytes + (STACK ? 4 * 1024 * 4 : 0) + 64;
  (void)hipFuncSetAttribute((const void*)chase<STACK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  unsigned long long h[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((chase<STACK>), dim3(cus), dim3(threads), ldsBytes, 0, tableBytes, lanes, hops, out, clocks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      (void)hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost);
    }
  }
  const double mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  const double clkPerHop = (double)h[0] / hops;  // shader clocks of workgroup 0's first wave (includes the table fill)
  const int waves = threads / 64;
  printf("LDS chase%s table %3u KB lanes %2d waves/CU %2d: %8.3f ms  %7.1f clk/hop/wave  %6.1f clk/hop/CU  (%.0f MHz)\n", STACK ? " + stack" : "        ",
         tableBytes >> 10, lanes, waves, best, clkPerHop, clkPerHop / waves, mhz);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  uint32_t* out;
  unsigned long long* clocks;
  hipMalloc(&out, (size_t)cus * 1024 * 4);
  hipMalloc(&clocks, 16);
  for (uint32_t bytes : {32u << 10, 128u << 10}) {
    for (int threads : {1024, 512}) {
      for (int lanes : {64, 38, 16}) {
        run<false>(bytes, lanes, threads, cus, out, clocks);
        run<true>(bytes, lanes, threads, cus, out, clocks);
      }
    }
  }
  return 0;
}
