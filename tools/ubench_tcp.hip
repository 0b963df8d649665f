// What does one divergent vector-memory instruction cost the CU's L1 (TCP) path when the data is cache
// resident?  The node step of the render kernel issues two buffer_load_dwordx4 per visit on ~38 of 64 lanes,
// each lane at its own 32-byte record of a 129 KB node array (TCP hit rate 95 %); a probe build showed that a
// third load per visit costs 11.7 % of the frame whether it is 4 or 16 bytes wide.  This bench issues
// INDEPENDENT loads (addresses do not depend on loaded data, many in flight) so the rate it reports is the
// path's throughput, not its latency:
//   hipcc --offload-arch=gfx950 -O3 -o ubench_tcp tools/ubench_tcp.hip && ./ubench_tcp
// Output: per variant (bytes per lane per record, instructions per record, active lanes, table size) the
// wave-instructions per microsecond per CU and the cycles per wave-instruction at the measured shader clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __amdgpu_buffer_rsrc_t Rsrc;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ Rsrc rsrcOf(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// WIDTH: 4 or 16 bytes per instruction; NINSTR: instructions per record (consecutive 16-byte pieces)
template <int WIDTH, int NINSTR>
__global__ __launch_bounds__(256) void gather(const void* table, uint32_t tableBytes, uint32_t recordBytes, int activeLanes, int perLane,
                                              uint32_t* out, unsigned long long* clocks) {
  const Rsrc rs = rsrcOf(table, tableBytes);
  const uint32_t records = tableBytes / recordBytes;  // power of two
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t acc = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (lane < activeLanes) {
    uint32_t x = i * 0x9E3779B1u;
#pragma unroll 4
    for (int k = 0; k < perLane; ++k) {
      x = x * 0x2545F491u + 0x3C6EF372u;  // LCG: address stream independent of the loaded data
      const uint32_t rec = (x >> 7) & (records - 1);
      const int off = (int)(rec * recordBytes);
#pragma unroll
      for (int j = 0; j < NINSTR; ++j) {
        if (WIDTH == 16) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16 * j, 0, 0);
          acc += v.x ^ v.w;
        } else {
          acc += __builtin_amdgcn_raw_buffer_load_b32(rs, off + 16 * j, 0, 0);
        }
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[i] = acc;
  if (i == 0) {
    clocks[0] = c1 - c0;
    clocks[1] = r1 - r0;
  }
}

// Dependent chain: the next record's index is the first dword of the one just loaded (what a traversal does);
// NINSTR b128 loads per hop.  Reports cycles per hop per wave with `wavesPerSimd` waves resident.
template <int NINSTR>
__global__ __launch_bounds__(256) void chase(const void* table, uint32_t tableBytes, uint32_t recordBytes, int activeLanes, int hops,
                                             uint32_t* out, unsigned long long* clocks) {
  const Rsrc rs = rsrcOf(table, tableBytes);
  const uint32_t records = tableBytes / recordBytes;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t acc = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (lane < activeLanes) {
    uint32_t rec = (i * 0x9E3779B1u >> 7) & (records - 1);
    for (int k = 0; k < hops; ++k) {
      const int off = (int)(rec * recordBytes);
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
#pragma unroll
      for (int j = 1; j < NINSTR; ++j) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16 * j, 0, 0);
        acc += w.y;
      }
      rec = (v.x + acc) & (records - 1);  // acc is 0 (pad dwords are 0) but keeps the other loads on the chain
    }
    acc += rec;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[i] = acc;
  if (i == 0) {
    clocks[0] = c1 - c0;
    clocks[1] = r1 - r0;
  }
}

template <int NINSTR>
static void runChase(const void* table, uint32_t tableBytes, uint32_t recordBytes, int activeLanes, int wgPerCu, uint32_t* out,
                     unsigned long long* clocks, int cus) {
  const int grid = cus * wgPerCu, hops = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  unsigned long long h[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((chase<NINSTR>), dim3(grid), dim3(256), 0, 0, table, tableBytes, recordBytes, activeLanes, hops, out, clocks);
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
  const double clkPerHop = best * 1e-3 * mhz * 1e6 / hops;  // every wave does `hops` hops in `best` ms
  printf("chase %d x b128  table %6u KB record %3u B lanes %2d waves/SIMD %d: %8.3f ms  %7.1f clk/hop/wave  %6.1f clk/hop/CU (TCP occupancy: %d waves)\n",
         NINSTR, tableBytes >> 10, recordBytes, activeLanes, wgPerCu, best, clkPerHop, clkPerHop / (4.0 * wgPerCu), 4 * wgPerCu);
}

template <int WIDTH, int NINSTR>
static void run(const char* name, const void* table, uint32_t tableBytes, uint32_t recordBytes, int activeLanes, uint32_t* out,
                unsigned long long* clocks, int cus) {
  const int grid = cus * 8, perLane = 2048;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  unsigned long long h[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((gather<WIDTH, NINSTR>), dim3(grid), dim3(256), 0, 0, table, tableBytes, recordBytes, activeLanes, perLane, out,
                       clocks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      (void)hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost);
    }
  }
  const double waveInstr = (double)grid * 4 * perLane * NINSTR;
  const double mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;  // s_memrealtime ticks at 100 MHz
  const double perUsCu = waveInstr / (best * 1e3) / cus;
  printf("%-10s table %6u KB record %3u B lanes %2d: %8.3f ms  %7.2f wave-instr/us/CU  %6.1f clk/wave-instr at %.0f MHz  (%.2f TB/s useful)\n",
         name, tableBytes >> 10, recordBytes, activeLanes, best, perUsCu, mhz / perUsCu, mhz,
         waveInstr * activeLanes * WIDTH / (best * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const uint32_t maxBytes = 64u << 20;
  void* table;
  uint32_t* out;
  unsigned long long* clocks;
  hipMalloc(&table, maxBytes);
  hipMemset(table, 1, maxBytes);
  hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
  hipMalloc(&clocks, 16);
  printf("%d CUs\n", cus);
  const uint32_t sizes[] = {16u << 10, 128u << 10, 2u << 20, 64u << 20};
  for (uint32_t bytes : sizes) {
    for (int lanes : {64, 38, 16}) {
      run<4, 1>("1 x b32", table, bytes, 32, lanes, out, clocks, cus);
      run<16, 1>("1 x b128", table, bytes, 32, lanes, out, clocks, cus);
      run<16, 2>("2 x b128", table, bytes, 32, lanes, out, clocks, cus);
      run<16, 3>("3 x b128", table, bytes, 64, lanes, out, clocks, cus);
      run<16, 4>("4 x b128", table, bytes, 64, lanes, out, clocks, cus);
    }
  }
  // chain tables: first dword of every record = a pseudo-random next index, the rest 0
  {
    const uint32_t n = maxBytes / 4;
    uint32_t* host = (uint32_t*)calloc(n, 4);
    uint32_t x = 12345u;
    for (uint32_t k = 0; k < n; k += 8) {  // 32-byte granularity serves the 32- and 64-byte records alike
      x = x * 0x2545F491u + 0x3C6EF372u;
      host[k] = x >> 5;
    }
    (void)hipMemcpy(table, host, maxBytes, hipMemcpyHostToDevice);
    free(host);
  }
  for (uint32_t bytes : sizes) {
    for (int wg : {5, 8, 2}) {
      for (int lanes : {64, 38}) {
        runChase<1>(table, bytes, 32, lanes, wg, out, clocks, cus);
        runChase<2>(table, bytes, 32, lanes, wg, out, clocks, cus);
        runChase<4>(table, bytes, 64, lanes, wg, out, clocks, cus);
      }
    }
  }
  return 0;
}
