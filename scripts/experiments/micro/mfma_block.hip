// MFMA experiment: the cost of applying one dense 16x16
// COMPLEX block (a round's 4-bit register window) to a wave's 64 x 16 amplitudes with
// v_mfma_f32_32x32x2_f32, next to the same wave doing a round's worth of the factored packed-fp32 gate
// math the engine uses today.  Developer tool:
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_block.hip -o scripts/micro/mfma_block.bin
//   gpurun -- ./scripts/micro/mfma_block.bin
//
// Real form of the block: [[Ur, -Ui], [Ui, Ur]] (32 x 32) times a column [re; im] (32).  One
// 32x32x2 MFMA multiplies 32 columns by two of the 32 k-indices; a column's 16 complex values sit in
// a lane PAIR (l, l + 32), so a wave (64 columns, one per lane) does: 16 v_permlane32_swap to split
// the columns over the pairs, 2 x 16 MFMAs (two groups of 32 columns), 16 swaps back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kRounds = 2048;

__global__ __launch_bounds__(256) void mfma_rounds(const float* __restrict__ umat, float* __restrict__ out, float seed) {
  const int lane = threadIdx.x & 63;
  // A operand of k-step s: A[row = lane % 32][k = 2 s + lane / 32] -- 16 registers hold the whole block
  float a[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) a[s] = umat[(lane & 31) * 32 + 2 * s + (lane >> 5)];
  float x[32];  // this lane's column: 16 complex amplitudes
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = seed + 0.001f * float(i + lane);
  for (int r = 0; r < kRounds; ++r) {
    // split: low lanes keep components 0..15 of their own column and receive 0..15 of the partner's
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[16 + i]), __float_as_uint(x[i]), false, false);
      x[16 + i] = __uint_as_float(sw[0]);
      x[i] = __uint_as_float(sw[1]);
    }
    v16f d0 = {0}, d1 = {0};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], x[s], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], x[16 + s], d1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = d0[i]; x[16 + i] = d1[i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[16 + i]), __float_as_uint(x[i]), false, false);
      x[16 + i] = __uint_as_float(sw[0]);
      x[i] = __uint_as_float(sw[1]);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) acc += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// The same wave doing what a config-3 forward round does on its 16 amplitudes per lane with the
// factored gates: 7 X**t (4 packed ops per amplitude pair), 2 FULL diagonal tables (2 per amplitude),
// 3 boundary phases on half the amplitudes -- 21 packed ops per amplitude and round.
__global__ __launch_bounds__(256) void valu_rounds(float* __restrict__ out, float seed) {
  v2f x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = v2f{seed + 0.001f * float(i), seed - 0.002f * float(threadIdx.x & 63)};
  const v2f cs = v2f{0.999f, 0.04f};
  for (int r = 0; r < kRounds; ++r) {
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      const int b = 1 << (g & 3);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i & b) continue;
        v2f t0, t1;
        asm volatile("v_pk_mul_f32 %[t0], %[a0], %[cs] op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
                     "v_pk_mul_f32 %[t1], %[a1], %[cs] op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
                     "v_pk_fma_f32 %[a0], %[a0], %[cs], %[t1] op_sel_hi:[1,0,1]\n\t"
                     "v_pk_fma_f32 %[a1], %[a1], %[cs], %[t0] op_sel_hi:[1,0,1]"
                     : [a0] "+v"(x[i]), [a1] "+v"(x[i | b]), [t0] "=&v"(t0), [t1] "=&v"(t1) : [cs] "v"(cs));
      }
    }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int i = 1; i < 16; ++i) {
        v2f t;
        asm volatile("v_pk_mul_f32 %[t], %[a], %[cs] op_sel_hi:[1,0]\n\t"
                     "v_pk_fma_f32 %[a], %[a], %[cs], %[t] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
                     : [a] "+v"(x[i]), [t] "=&v"(t) : [cs] "v"(cs));
      }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (!(i & (1 << c))) continue;
        v2f t;
        asm volatile("v_pk_mul_f32 %[t], %[a], %[cs] op_sel_hi:[1,0]\n\t"
                     "v_pk_fma_f32 %[a], %[a], %[cs], %[t] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
                     : [a] "+v"(x[i]), [t] "=&v"(t) : [cs] "v"(cs));
      }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<float> h(32 * 32);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) h[i * 32 + j] = (i == j) ? 0.9999f : 1e-4f * float((i * 7 + j) % 5 - 2);
  float *umat, *out;
  hipMalloc(&umat, h.size() * 4);
  hipMemcpy(umat, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, size_t(cus) * 16 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("device %s, %d CUs, clock %.2f GHz nominal\n", prop.name, cus, prop.clockRate * 1e-6);
  for (int waves_per_simd : {1, 2, 4}) {
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(mfma_rounds, dim3(blocks), dim3(256), 0, 0, umat, out, 0.5f);
        else hipLaunchKernelGGL(valu_rounds, dim3(blocks), dim3(256), 0, 0, out, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double amp_rounds = double(blocks) * 256.0 * 16.0 * kRounds;
      // SIMD-cycles per amplitude and round at the nominal clock: time x clock x SIMDs / amplitude-rounds
      const double cyc = ms * 1e-3 * prop.clockRate * 1e3 * (cus * 4.0) / amp_rounds;
      printf("%-28s %d wave(s)/SIMD: %8.3f ms  -> %.2f SIMD-cycles per amplitude and round (nominal clock)\n",
             which == 0 ? "dense 16x16 complex via MFMA" : "factored gates via v_pk_*", waves_per_simd, ms, cyc);
    }
  }
  return 0;
}
