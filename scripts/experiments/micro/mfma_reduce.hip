// Round 6 (review item 4 c): can the idle MATRIX pipe take over the wave reductions of the adjoint kernel?
//
// The adjoint pass reduces up to eight gradient partials over the 64 lanes of a wave per slot group (kernels.hip
// add_slots8: 12 DPP adds, v_permlane16_swap, two quad butterflies, v_permlane32_swap; 13.5 % of the kernel's VALU
// instructions) while the MFMA unit idles.  A matrix instruction with a ones operand is a reduction in a fixed hardware
// order (bit-reproducible), so the experiment: eight vectors x 64 lanes -> eight sums
//   DPP        the shipped chain
//   MFMA       per vector  D = A(16x4: the vector) x ones(4x16)  (v_mfma_f32_16x16x4_f32: sums lanes l, l+16, l+32, l+48),
//              three v_add_f32 over D's four registers, and  ones(16x4) x E(4x16)  for the remaining 16 -> 1:
//              16 matrix instructions + 24 VALU adds per call.  (An operand holds 64 values, so eight vectors need at
//              least eight matrix instructions of 8 passes each whatever the shape: v_mfma_f32_4x4x1_16B has k = 1 and
//              reduces nothing across lanes, 32x32x2 reduces pairs in 16 passes.)
// alone and interleaved with FILL packed FMAs per call (the instance arithmetic the reduction would hide under), at four
// waves per SIMD.  Reported: shader cycles per call and wave (s_memtime), and per call and SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/micro/mfma_reduce.hip -o mfma_reduce.bin && ./mfma_reduce.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kIters = 2048;

__device__ __forceinline__ float reduce8_dpp(float g0, float g1, float g2, float g3, float g4, float g5, float g6, float g7) {
  float t0, t1, t2, t3, u0, u1, w, x;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %[t0], %[g0], %[g0] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %[t1], %[g2], %[g2] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %[t2], %[g4], %[g4] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %[t3], %[g6], %[g6] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %[t0], %[g1], %[g1] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %[t1], %[g3], %[g3] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %[t2], %[g5], %[g5] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %[t3], %[g7], %[g7] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %[u0], %[t0], %[t0] row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %[u1], %[t2], %[t2] row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %[u0], %[t1], %[t1] row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %[u1], %[t3], %[t3] row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %[u0], %[u1]\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %[w], %[u0], %[u1]\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %[w], %[w], %[w] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %[w], %[w], %[w] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32 %[x], %[w]\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %[w], %[x]\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %[w], %[w], %[x]"
      : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [w] "=&v"(w), [x] "=&v"(x)
      : [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [g3] "v"(g3), [g4] "v"(g4), [g5] "v"(g5), [g6] "v"(g6), [g7] "v"(g7));
  return w;
}

// One vector -> its sum in every lane, through the matrix pipe.
__device__ __forceinline__ float reduce1_mfma(float g) {
  const v4f zero = {0.f, 0.f, 0.f, 0.f};
  const v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(g, 1.0f, zero, 0, 0, 0);     // D[i][j] = sum_k g[lane i + 16 k]
  const float e = (d.x + d.y) + (d.z + d.w);                                        // rows 4 q .. 4 q + 3 of lane group q
  const v4f f = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, e, zero, 0, 0, 0);      // sum over the four lane groups
  return f.x;
}

template <int MODE, int FILL>
__global__ __launch_bounds__(1024) void k(float* out, uint64_t* cyc, float x) {
  float g[8];
  for (int i = 0; i < 8; ++i) g[i] = x * float(threadIdx.x + i);
  v2f a0{x, 1.f}, a1{x, 2.f}, a2{x, 3.f}, a3{x, 4.f};
  const v2f c{1.0001f, 0.0001f};
  float acc = 0.f;
  uint64_t t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int f = 0; f < FILL / 4; ++f)
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));
    float r;
    if (MODE == 0) r = reduce8_dpp(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
    else if (MODE == 1) {
      r = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) r += reduce1_mfma(g[i]);
    } else r = 0.f;
    acc += r;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] += 1e-9f * r;   // (the next call depends on this one, as an instance's partials do not: a bound)
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) cyc[gw] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + a0.x + a1.x + a2.x + a3.x;
}

template <int MODE, int FILL>
double run(float* d, uint64_t* dc, int n_cus) {
  hipLaunchKernelGGL((k<MODE, FILL>), dim3(n_cus), dim3(1024), 0, 0, d, dc, 1.0f);
  hipLaunchKernelGGL((k<MODE, FILL>), dim3(n_cus), dim3(1024), 0, 0, d, dc, 1.0f);
  (void)hipDeviceSynchronize();
  std::vector<uint64_t> h(size_t(n_cus) * 16);
  (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (uint64_t v : h) s += double(v);
  return s / double(h.size()) / kIters;   // cycles per call and wave (four waves share a SIMD)
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int n_cus = prop.multiProcessorCount;
  float* d; uint64_t* dc;
  (void)hipMalloc(&d, size_t(n_cus) * 1024 * 4); (void)hipMalloc(&dc, size_t(n_cus) * 16 * 8);
  printf("eight 64-lane vectors -> eight sums, %d CUs, four waves per SIMD; shader cycles per call and WAVE (per call and SIMD = / 4)\n", n_cus);
  const double base0 = run<2, 0>(d, dc, n_cus), base64 = run<2, 64>(d, dc, n_cus), base128 = run<2, 128>(d, dc, n_cus);
  printf("%-44s %8.1f %8.1f %8.1f\n", "loop + FILL packed FMAs alone (0 / 64 / 128)", base0, base64, base128);
  const double d0 = run<0, 0>(d, dc, n_cus), d64 = run<0, 64>(d, dc, n_cus), d128 = run<0, 128>(d, dc, n_cus);
  printf("%-44s %8.1f %8.1f %8.1f   added by the reduction: %6.1f %6.1f %6.1f\n", "DPP chain (shipped) with FILL 0 / 64 / 128", d0, d64, d128,
         d0 - base0, d64 - base64, d128 - base128);
  const double m0 = run<1, 0>(d, dc, n_cus), m64 = run<1, 64>(d, dc, n_cus), m128 = run<1, 128>(d, dc, n_cus);
  printf("%-44s %8.1f %8.1f %8.1f   added by the reduction: %6.1f %6.1f %6.1f\n", "16 x v_mfma_f32_16x16x4_f32 + 24 v_add_f32", m0, m64, m128,
         m0 - base0, m64 - base64, m128 - base128);
  return 0;
}
