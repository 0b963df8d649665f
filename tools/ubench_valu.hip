// Issue-rate microbenchmark for the VALU instruction classes the shading step uses (gfx950):
// cycles per wave-instruction, one wave per SIMD and five waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void k(float* out, int iters, long long* cyc) {
  float a0 = threadIdx.x * 1.0001f + 1.0f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
  uint32_t u0 = threadIdx.x * 2654435761u + 1u, u1 = u0 ^ 0x9e3779b9u, u2 = u0 + 77u, u3 = u0 * 3u;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 2) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 3) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 4) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 5) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 6) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 7) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_lshlrev_b32 %1, 13, %1\n v_alignbit_b32 %2, %2, %3, 7\n v_add_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 8) { REP16(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %0\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
    if (OP == 9) { REP16(asm volatile("v_mul_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_floor_f64 %2, %2\n v_cvt_f64_f32 %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0));) }
    if (OP == 10) { unsigned long long w0 = u0, w1 = u1; REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n v_mad_u64_u32 %0, vcc, %2, %2, %0\n v_mad_u64_u32 %1, vcc, %3, %3, %1" : "+v"(w0), "+v"(w1) : "v"(u2), "v"(u3) : "vcc");) u0 ^= (uint32_t)w0 ^ (uint32_t)(w1 >> 32); }
    if (OP == 11) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0" : "+v"(d0), "+v"(d1));) }
    if (OP == 13) { REP16(asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 14) { REP16(asm volatile("v_min3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 15) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 16) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 17) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
    if (OP == 18) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
    if (OP == 19) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 20) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(d0), "+v"(d1));) }
    if (OP == 21) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_min_f32 %1, %1, %2\n v_fma_f32 %2, %2, %3, %0\n v_max_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 22) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_fma_f32 %1, %1, %2, %3\n s_nop 0\n v_fma_f32 %2, %2, %3, %0\n s_nop 0\n v_fma_f32 %3, %3, %0, %1\n s_nop 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 23) { REP16(asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 24) { REP16(asm volatile("v_fma_f32 %0, %4, %1, %2\n v_fma_f32 %1, %4, %2, %3\n v_fma_f32 %2, %4, %3, %0\n v_fma_f32 %3, %4, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(iters));) }
    if (OP == 12) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %1, %2\n v_min3_f32 %2, %2, %3, %0\n v_max_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(d0 + d1 + d2 + d3);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, long long* cyc, int blocksPerCU) {
  const int iters = 2000, cus = 256;
  const int grid = cus * blocksPerCU;
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  const double instr = (double)iters * 64.0;  // per wave
  // SIMD cycles per wave-instruction = wall cycles / (instructions per wave x waves per SIMD)
  printf("%-26s %d wave(s)/SIMD: %6.2f clk/instr per wave (clock64), %6.2f SIMD-cycles/instr (at 2.4 GHz from %.3f ms)\n", name, blocksPerCU,
         (double)h[0] / instr, ms * 1e-3 * 2.4e9 / (instr * blocksPerCU), ms);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * 8 * 8);
  for (int b : {1, 4, 5, 8}) {
#define R(op, name) run<op>(name, out, cyc, b);
    R(13, "v_min_f32") R(14, "v_min3/max3_f32") R(15, "v_add_f32") R(16, "v_mul_f32") R(17, "v_cmp_lt_f32 -> vcc") R(18, "v_cndmask_b32") R(19, "v_add_u32")
    R(20, "v_pk_add/mul_f32") R(21, "fma/min/fma/max") R(22, "v_fma_f32 + s_nop 0") R(23, "v_med3_f32") R(24, "v_fma_f32 (one SGPR operand)")
    R(0, "v_fma_f32") R(12, "cndmask/cmp/min3/max") R(7, "xor/shift/alignbit/add") R(1, "v_mul_lo_u32") R(2, "v_mul_hi_u32") R(10, "v_mad_u64_u32")
    R(6, "v_mul_u32_u24") R(3, "v_fma_f64") R(9, "mul/add/floor/cvt f64") R(4, "v_rcp_f32") R(5, "v_sqrt_f32") R(8, "div_scale/fmas/fixup/cvt") R(11, "v_pk_fma_f32")
  }
  return 0;
}
