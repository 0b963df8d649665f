// Calibration of rocprofv3's FETCH_SIZE for the traversal's access shape (MI355X_MICROARCH.md, HBM: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern"):
// every lane reads random 32-byte records (two dwordx4 loads, like a BVH node visit) from a table far larger
// than L2 + Infinity Cache, each record once per pass.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_gather tools/ubench_gather.hip
//   ./ubench_gather                      -> records/s, GB/s at 32 / 64 / 128 B per record
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./ubench_gather   -> FETCH_SIZE per launch
// If records/s x 128 B exceeds the 8 TB/s peak the memory system cannot be fetching 128 B per record, and
// FETCH_SIZE / records tells which granularity the counter tallies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void gather(const float4* table, uint32_t records, uint32_t mult, float* out, int perLane) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
  const uint32_t total = gridDim.x * blockDim.x;
  for (int k = 0; k < perLane; ++k) {
    // bijection on [0, records) for records = 2^n: odd multiplier, then xor-shift; consecutive lanes land far apart
    uint32_t x = (i + (uint32_t)k * total) & (records - 1);
    x = (x * mult) & (records - 1);
    x ^= x >> 13;
    x = (x * 0x9E3779B1u) & (records - 1);
    const float4 a = table[2 * (size_t)x], b = table[2 * (size_t)x + 1];
    acc += a.x + b.w;
  }
  out[i] = acc;
}

int main() {
  const uint32_t records = 1u << 26;  // 2 GiB of 32-byte records
  float4* table;
  float* out;
  hipMalloc(&table, (size_t)records * 32);
  hipMemset(table, 0, (size_t)records * 32);
  const int grid = 256 * 8, perLane = 64;  // 2^19 lanes x 64 records = 2^25 records per launch (half the table)
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, table, records, 0x2545F491u + 2u * rep, out, perLane);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)grid * 256 * perLane;
    printf("launch %d: %.0f records in %.3f ms = %.2f G records/s; at 32 / 64 / 128 B per record: %.0f / %.0f / %.0f GB/s\n", rep, n, ms,
           n / ms / 1e6, n * 32 / ms / 1e6, n * 64 / ms / 1e6, n * 128 / ms / 1e6);
  }
  return 0;
}
