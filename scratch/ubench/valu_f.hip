// VALU issue-rate microbenchmark, float side: cycles per wave64 instruction per SIMD for the float / double /
// packed-float / conversion instructions the detector's score kernel is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float v[16];
  unsigned long long msk = __ballot(threadIdx.x & 1);
  double d[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i * a;
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = v[i], p[i] = f2{v[i], v[i + 8]};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
      if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
      if (OP == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
      if (OP == 4) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(v[i]));
      if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
      if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b));
      if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 20) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(b) : "vcc");
      if (OP == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "s"(msk));
      if (OP == 22) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 23) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b) : "vcc");
      if (OP == 24) { unsigned long long m2; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(v[i]), "v"(b)); asm volatile("" :: "s"(m2)); }
      if (OP == 25) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(v[i]) : "v"(b));
      if (OP == 26) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      // two of each per i, so every OP issues 16 instructions per iteration
      if (OP == 10) asm volatile("v_add_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)b));
      if (OP == 11) asm volatile("v_cvt_f64_f32 %0, %1\n\tv_cvt_f64_f32 %0, %2" : "=&v"(d[i]) : "v"(v[i]), "v"(v[i + 8]));
      if (OP == 12) asm volatile("v_cvt_f32_f64 %0, %2\n\tv_cvt_f32_f64 %1, %2" : "=&v"(v[i]), "=&v"(v[i + 8]) : "v"(d[i]));
      if (OP == 13) asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f2{a, b}));
      if (OP == 14) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f2{a, b}));
      if (OP == 15) asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(f2{a, b}));
      if (OP == 16) asm volatile("v_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"((double)b));
      if (OP == 17) asm volatile("v_mul_f64 %0, %0, %1\n\tv_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)a));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; i++) s += (float)d[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int blocks_per_cu) {
  float* d; hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
  hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)iters * 16 * blocks_per_cu;  // each block = 4 waves, one per SIMD
  printf("%-18s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_add_f32", w); run<1>("v_mul_f32", w); run<2>("v_fma_f32", w); run<3>("v_sqrt_f32", w); run<4>("v_cvt_f32_ubyte0", w);
    run<5>("v_max_f32", w); run<6>("v_max3_f32", w); run<7>("v_cndmask_b32", w); run<8>("v_rcp_f32", w); run<20>("v_cmp vcc", w); run<21>("v_cndmask_e64 sgpr", w); run<22>("v_and_b32", w); run<23>("v_cmp+v_cndmask", w); run<24>("v_cmp_e64 sgpr", w); run<25>("v_cndmask swapped", w); run<26>("v_cndmask indep", w);
    run<10>("v_add_f64", w); run<11>("v_cvt_f64_f32", w); run<12>("v_cvt_f32_f64", w); run<13>("v_pk_add_f32", w);
    run<14>("v_pk_mul_f32", w); run<15>("v_pk_fma_f32", w); run<16>("v_fma_f64", w); run<17>("v_mul_f64", w);
  }
  return 0;
}
