// Does v_dot2_u32_u16 / v_dot2c_i32_i16 read both halves of an SGPR source?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned int a, unsigned int w, unsigned int* out) {
  unsigned int av = a + threadIdx.x * 0;  // VGPR
  unsigned int wv = w + threadIdx.x * 0;  // VGPR copy
  unsigned int ws;
  asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ws) : "v"(wv));
  unsigned int r0, r1, r2 = 5, r3 = 5, zero = threadIdx.x * 0;
  asm volatile("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r0) : "v"(av), "v"(wv), "v"(zero));
  asm volatile("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r1) : "v"(av), "s"(ws), "v"(zero));
  asm volatile("v_dot2c_i32_i16_e32 %0, %1, %2" : "+v"(r2) : "v"(wv), "v"(av));
  asm volatile("v_dot2c_i32_i16_e32 %0, %1, %2" : "+v"(r3) : "s"(ws), "v"(av));
  if (threadIdx.x == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; }
}
int main() {
  unsigned int* d; hipMalloc(&d, 16);
  unsigned int a = 3u | (7u << 16), w = 100u | (1000u << 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, w, d);
  unsigned int h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("expect %u: dot2_u32_u16 vgpr %u sgpr %u ; dot2c vgpr %u sgpr %u (expect %u)\n", 3 * 100 + 7 * 1000, h[0], h[1], h[2], h[3], 5 + 7300);
  return 0;
}
