// Issue-rate microbenchmark (developer tool): cycles per wave instruction of v_fma_f32,
// v_pk_fma_f32, v_pk_mul_f32, v_readlane_b32 and v_mov_b32 at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;

template <int MODE>
__global__ void k(float* out, float x) {
  v2f a0{x, 1.f}, a1{x, 2.f}, a2{x, 3.f}, a3{x, 4.f}, a4{x, 5.f}, a5{x, 6.f}, a6{x, 7.f}, a7{x, 8.f};
  v2f c{1.0001f, 0.0001f};
  int s0 = 0, s1 = 0;
  for (int i = 0; i < kIters; ++i) {
    if (MODE == 0) {  // 16 scalar fma (independent)
      asm volatile(
          "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
          "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
          "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
          "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
          : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(c.x));
    } else if (MODE == 1) {  // 16 pk_fma
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
          "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
          "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 2) {  // 16 pk_mul
      asm volatile(
          "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
          "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
          "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
          "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 3) {  // 16 readlane
      asm volatile(
          "v_readlane_b32 %0, %2, 1\n v_readlane_b32 %1, %2, 2\n v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %2, 4\n"
          "v_readlane_b32 %0, %2, 5\n v_readlane_b32 %1, %2, 6\n v_readlane_b32 %0, %2, 7\n v_readlane_b32 %1, %2, 8\n"
          "v_readlane_b32 %0, %2, 9\n v_readlane_b32 %1, %2, 10\n v_readlane_b32 %0, %2, 11\n v_readlane_b32 %1, %2, 12\n"
          "v_readlane_b32 %0, %2, 13\n v_readlane_b32 %1, %2, 14\n v_readlane_b32 %0, %2, 15\n v_readlane_b32 %1, %2, 16\n"
          : "=s"(s0), "=s"(s1) : "v"(a0.x));
    } else if (MODE == 4) {  // 16 v_mov_b32
      asm volatile(
          "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
          "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
          "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
          "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
          : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));
    } else if (MODE == 5) {  // pk_fma with SGPR operand + op_sel modifiers (as in the engine)
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %1, %1, %8, %2 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %2, %8, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %3, %3, %8, %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %4, %4, %8, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %5, %5, %8, %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %6, %8, %7 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %7, %7, %8, %0 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %0, %0, %8, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %1, %1, %8, %2 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %2, %2, %8, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %3, %3, %8, %4 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %4, %4, %8, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %5, %5, %8, %6 op_sel_hi:[1,0,1]\n"
          "v_pk_fma_f32 %6, %6, %8, %7 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %7, %7, %8, %0 op_sel_hi:[1,0,1]\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a7.y + float(s0 + s1);
}

template <int MODE>
void run(const char* name, float* d) {
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps;  // 4 SIMDs x wps waves
    const int blocks = 256;         // one workgroup per CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = double(kIters) * 16 * wps;
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_pk_mul_f32", d);
  run<3>("v_readlane_b32", d); run<4>("v_mov_b32", d); run<5>("v_pk_fma_f32 sgpr+op_sel", d);
  return 0;
}
