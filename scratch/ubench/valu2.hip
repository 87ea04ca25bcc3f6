// Issue cost of single vector instructions on gfx950, in SHADER CYCLES (s_memtime deltas inside the kernel, not
// wall time at an assumed clock): W waves per SIMD each run a loop of 32 copies of one instruction whose destinations rotate over four registers (two for the
// float forms), so that consecutive copies do not wait on each other's write.
//   hipcc -O3 --offload-arch=gfx950 scratch/ubench/valu2.hip -o scratch/ubench/valu2 && scratch/ubench/valu2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X)  // each X holds four copies with rotating destination registers

#define KERNEL(NAME, ASM)                                                                           \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters, int a, float f) { \
    int v0 = threadIdx.x + a, v1 = threadIdx.x * 3 + a, v2 = a + 7, v3 = threadIdx.x ^ a;           \
    float f0 = f + threadIdx.x, f1 = f * 3.f;                                                       \
    double d0 = f0, d1 = f1;                                                                        \
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;                                                             \
    float g0 = 0, g1 = 0;                                                                           \
    double e0 = 0;                                                                                  \
    int s0 = 0;                                                                                     \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int it = 0; it < iters; it++) {                                                            \
      REP32(asm volatile(ASM                                                                        \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(g0), "+v"(g1), "+v"(e0), "+s"(s0) \
                         : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(f0), "v"(f1), "v"(d0), "v"(d1)    \
                         : "vcc", "scc");)                                                          \
    }                                                                                               \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                \
    if (r0 + r1 + r2 + r3 + (int)g0 + (int)g1 + (int)e0 + s0 == 0x7fffffff) out[0] = 0;             \
  }

// operands: %0..%3 int dst, %4 %5 float dst, %6 double dst, %7 sgpr dst; %8..%11 int src, %12 %13 float src, %14 %15 double src
KERNEL(k_add, "v_add_u32 %0, %8, %9\nv_add_u32 %1, %8, %9\nv_add_u32 %2, %8, %9\nv_add_u32 %3, %8, %9\n")
KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %8, %9\nv_dot2c_i32_i16 %1, %8, %9\nv_dot2c_i32_i16 %2, %8, %9\nv_dot2c_i32_i16 %3, %8, %9\n")
KERNEL(k_dot2_3op, "v_dot2_i32_i16 %0, %8, %9, %10\nv_dot2_i32_i16 %1, %8, %9, %10\nv_dot2_i32_i16 %2, %8, %9, %10\nv_dot2_i32_i16 %3, %8, %9, %10\n")
KERNEL(k_udot2, "v_dot2_u32_u16 %0, %8, %9, %10\nv_dot2_u32_u16 %1, %8, %9, %10\nv_dot2_u32_u16 %2, %8, %9, %10\nv_dot2_u32_u16 %3, %8, %9, %10\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %8, 9, 16\nv_bfe_u32 %1, %8, 9, 16\nv_bfe_u32 %2, %8, 9, 16\nv_bfe_u32 %3, %8, 9, 16\n")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 9, %8\nv_ashrrev_i32 %1, 9, %8\nv_ashrrev_i32 %2, 9, %8\nv_ashrrev_i32 %3, 9, %8\n")
KERNEL(k_sdwa, "v_ashrrev_i32_sdwa %0, 9, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\nv_ashrrev_i32_sdwa %1, 9, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\nv_ashrrev_i32_sdwa %2, 9, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\nv_ashrrev_i32_sdwa %3, 9, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n")
KERNEL(k_perm, "v_perm_b32 %0, %8, %9, %10\nv_perm_b32 %1, %8, %9, %10\nv_perm_b32 %2, %8, %9, %10\nv_perm_b32 %3, %8, %9, %10\n")
KERNEL(k_dpp, "v_add_u32_dpp %0, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_add_u32_dpp %1, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_add_u32_dpp %2, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_add_u32_dpp %3, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(k_rfl, "v_readfirstlane_b32 %7, %8\nv_readfirstlane_b32 %7, %8\nv_readfirstlane_b32 %7, %8\nv_readfirstlane_b32 %7, %8\n")
KERNEL(k_mulf, "v_mul_f32 %4, %12, %13\nv_mul_f32 %5, %12, %13\nv_mul_f32 %4, %12, %13\nv_mul_f32 %5, %12, %13\n")
KERNEL(k_fma, "v_fma_f32 %4, %12, %13, %12\nv_fma_f32 %5, %12, %13, %12\nv_fma_f32 %4, %12, %13, %12\nv_fma_f32 %5, %12, %13, %12\n")
KERNEL(k_pkmul, "v_pk_mul_f32 %6, %14, %15\nv_pk_mul_f32 %6, %14, %15\nv_pk_mul_f32 %6, %14, %15\nv_pk_mul_f32 %6, %14, %15\n")
KERNEL(k_pkadd, "v_pk_add_f32 %6, %14, %15\nv_pk_add_f32 %6, %14, %15\nv_pk_add_f32 %6, %14, %15\nv_pk_add_f32 %6, %14, %15\n")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %4, %8\nv_cvt_f32_i32 %5, %8\nv_cvt_f32_i32 %4, %8\nv_cvt_f32_i32 %5, %8\n")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %12\nv_cvt_i32_f32 %1, %12\nv_cvt_i32_f32 %2, %12\nv_cvt_i32_f32 %3, %12\n")
KERNEL(k_floor, "v_floor_f32 %4, %12\nv_floor_f32 %5, %12\nv_floor_f32 %4, %12\nv_floor_f32 %5, %12\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %12, %13\nv_cmp_lt_f32 vcc, %12, %13\nv_cmp_lt_f32 vcc, %12, %13\nv_cmp_lt_f32 vcc, %12, %13\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %8, %9, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cndmask_b32 %3, %8, %9, vcc\n")
KERNEL(k_mad24, "v_mad_i32_i24 %0, %8, %9, %10\nv_mad_i32_i24 %1, %8, %9, %10\nv_mad_i32_i24 %2, %8, %9, %10\nv_mad_i32_i24 %3, %8, %9, %10\n")
KERNEL(k_mul24, "v_mul_i32_i24 %0, %8, %9\nv_mul_i32_i24 %1, %8, %9\nv_mul_i32_i24 %2, %8, %9\nv_mul_i32_i24 %3, %8, %9\n")
KERNEL(k_and, "v_and_b32 %0, %8, %9\nv_and_b32 %1, %8, %9\nv_and_b32 %2, %8, %9\nv_and_b32 %3, %8, %9\n")
KERNEL(k_and_or, "v_and_or_b32 %0, %8, %9, %10\nv_and_or_b32 %1, %8, %9, %10\nv_and_or_b32 %2, %8, %9, %10\nv_and_or_b32 %3, %8, %9, %10\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %8, 3, %9\nv_lshl_add_u32 %1, %8, 3, %9\nv_lshl_add_u32 %2, %8, 3, %9\nv_lshl_add_u32 %3, %8, 3, %9\n")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %8, %9, %10\nv_alignbyte_b32 %1, %8, %9, %10\nv_alignbyte_b32 %2, %8, %9, %10\nv_alignbyte_b32 %3, %8, %9, %10\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\n")
KERNEL(k_cvt_f64_f32, "v_cvt_f64_f32 %6, %12\nv_cvt_f64_f32 %6, %12\nv_cvt_f64_f32 %6, %12\nv_cvt_f64_f32 %6, %12\n")
KERNEL(k_add_f64, "v_add_f64 %6, %14, %15\nv_add_f64 %6, %14, %15\nv_add_f64 %6, %14, %15\nv_add_f64 %6, %14, %15\n")
KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 %4, %14\nv_cvt_f32_f64 %5, %14\nv_cvt_f32_f64 %4, %14\nv_cvt_f32_f64 %5, %14\n")
KERNEL(k_sqrt, "v_sqrt_f32 %4, %12\nv_sqrt_f32 %5, %12\nv_sqrt_f32 %4, %12\nv_sqrt_f32 %5, %12\n")
KERNEL(k_snop, "s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n")
KERNEL(k_salu, "s_add_u32 %7, %7, 1\ns_add_u32 %7, %7, 1\ns_add_u32 %7, %7, 1\ns_add_u32 %7, %7, 1\n")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %8, %9, %10\nv_pk_mad_u16 %1, %8, %9, %10\nv_pk_mad_u16 %2, %8, %9, %10\nv_pk_mad_u16 %3, %8, %9, %10\n")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %8, %9\nv_pk_add_u16 %1, %8, %9\nv_pk_add_u16 %2, %8, %9\nv_pk_add_u16 %3, %8, %9\n")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %8, %9, %10\nv_mad_u32_u24 %1, %8, %9, %10\nv_mad_u32_u24 %2, %8, %9, %10\nv_mad_u32_u24 %3, %8, %9, %10\n")
KERNEL(k_sub_f32, "v_sub_f32 %4, %12, %13\nv_sub_f32 %5, %12, %13\nv_sub_f32 %4, %12, %13\nv_sub_f32 %5, %12, %13\n")
KERNEL(k_max_f32, "v_max_f32 %4, %12, %13\nv_max_f32 %5, %12, %13\nv_max_f32 %4, %12, %13\nv_max_f32 %5, %12, %13\n")

typedef void (*kfn)(unsigned long long*, int, int, float);
struct Entry { const char* name; kfn fn; };

int main() {
  Entry es[] = {{"v_add_u32", k_add}, {"v_dot2c_i32_i16", k_dot2c}, {"v_dot2_i32_i16 (3 op)", k_dot2_3op}, {"v_dot2_u32_u16", k_udot2},
                {"v_bfe_u32", k_bfe}, {"v_ashrrev_i32", k_ashr}, {"v_ashrrev_i32_sdwa", k_sdwa}, {"v_perm_b32", k_perm},
                {"v_add_u32_dpp", k_dpp}, {"v_readfirstlane_b32", k_rfl}, {"v_mul_f32", k_mulf}, {"v_fma_f32", k_fma},
                {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd}, {"v_cvt_f32_i32", k_cvt_f32_i32},
                {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_floor_f32", k_floor}, {"v_cmp_lt_f32", k_cmp}, {"v_cndmask_b32", k_cndmask},
                {"v_mad_i32_i24", k_mad24}, {"v_mul_i32_i24", k_mul24}, {"v_and_b32", k_and}, {"v_and_or_b32", k_and_or},
                {"v_lshl_add_u32", k_lshl_add}, {"v_alignbyte_b32", k_alignbyte}, {"v_mov_b32", k_mov},
                {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_add_f64", k_add_f64}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_sqrt_f32", k_sqrt},
                {"s_nop 0", k_snop}, {"s_add_u32", k_salu}, {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_add_u16", k_pk_add_u16},
                {"v_mad_u32_u24", k_mad_u32_u24}, {"v_sub_f32", k_sub_f32}, {"v_max_f32", k_max_f32}};
  setvbuf(stdout, NULL, _IONBF, 0);
  unsigned long long* d;
  hipMalloc(&d, 256 * 8 * 4 * 8);
  const int iters = 2000;
  printf("%-24s", "cycles per instruction per SIMD at W waves per SIMD:");
  printf("\n%-24s %8s %8s %8s %8s\n", "instruction", "W=1", "W=2", "W=4", "W=8");
  for (auto& e : es) {
    printf("%-24s", e.name);
    for (int w : {1, 2, 4, 8}) {
      const int blocks = 256 * w;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 50, 3, 1.5f);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 3, 1.5f);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks * 4);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2];
      // a SIMD hosts w waves (if the blocks spread evenly: 256 CUs x w blocks x 4 waves); each issues iters * 32
      printf(" %8.2f", med / ((double)iters * 32 * w));
    }
    printf("\n");
  }
  return 0;
}
