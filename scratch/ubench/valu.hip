// VALU issue-rate microbenchmark: how many cycles does one wave64 integer VALU instruction take per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters, int a, int b) {
  int v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i * a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == 0) v[i] = v[i] + b;                                  // v_add_u32
      if (OP == 1) v[i] = __mul24(v[i], a) + b;                      // v_mad_i32_i24
      if (OP == 2) v[i] = __builtin_amdgcn_perm(v[i], b, 0x0c010c00); // v_perm_b32
      if (OP == 3) v[i] = v[i] * a;                                  // v_mul_lo_u32
      if (OP == 4) { typedef short s2 __attribute__((ext_vector_type(2))); v[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, v[i]), __builtin_bit_cast(s2, b), v[i], false); }
      if (OP == 5) v[i] = v[i] + __builtin_amdgcn_update_dpp(0, v[i], 0xB1, 0xf, 0xf, false);
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int blocks_per_cu) {
  int* d; hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 3, 5);
  hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 3, 5); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)iters * 16 * blocks_per_cu;  // each block = 4 waves, one per SIMD
  printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_add_u32", w); run<1>("v_mad_i32_i24", w); run<2>("v_perm_b32", w); run<3>("v_mul_lo_u32", w); run<4>("v_dot2c_i32_i16", w); run<5>("v_add_dpp", w);
  }
  return 0;
}
