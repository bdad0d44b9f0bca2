// Issue cost of the VALU instructions the pass kernels are made of, in SHADER CYCLES (s_memtime) and
// with the sustained clock of the run next to it (s_memrealtime, 100 MHz) -- VERDICT r4 "what's weak" 4(b):
// valu_rate.hip converted wall time at a nominal 2.4 GHz, so a slow instruction and a throttled clock
// looked the same.  Developer tool, not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/micro/valu_cycles.hip -o valu_cycles.bin && ./valu_cycles.bin
//
// Every wave runs kIters iterations of a 16-instruction body on independent registers (dependent
// instructions 8 apart unless the mode says otherwise) between two s_memtime reads; one workgroup per CU,
// 1 / 2 / 4 waves per SIMD.  Reported: cycles per wave-instruction and SIMD = wave cycles / (kIters * 16 *
// waves per SIMD), the mean clock = shader cycles / real time, and the packed-fp32 rate that follows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 8192;

#define R8(T_) T_(0, 1) T_(1, 2) T_(2, 3) T_(3, 4) T_(4, 5) T_(5, 6) T_(6, 7) T_(7, 0)
#define R16(T_) R8(T_) R8(T_)

enum Mode {
  FMA_V = 0, FMA_S, FMAC_V, PKFMA_V, PKFMA_S, PKFMA_SEL, PKMUL_V, PKADD_V, MOV, XOR, CNDMASK, ADD_DPP, MUL_V, ADD_V,
  MIX_PK_FMA, MIX_PK_MOV, PKFMA_DEP1, PKFMA_DEP2, PKFMA_DEP4, SNOP_PK, PKFMA_3SRC, MOV64, FMA_3SRC, PERMLANE, CND_E64, CND_VCCSET, READLANE, WRITELANE, LSHL_ADD_U64, PKFMA_BANK, FMA_SS, N_MODES
};
const char* kNames[N_MODES] = {
    "v_fma_f32 (vgpr x3, 2 distinct)", "v_fma_f32 (one sgpr operand)", "v_fmac_f32 (VOP2, 4-byte)",
    "v_pk_fma_f32 (vgpr)", "v_pk_fma_f32 (sgpr operand)", "v_pk_fma_f32 (sgpr + op_sel/neg)", "v_pk_mul_f32 (vgpr)",
    "v_pk_add_f32 (vgpr)", "v_mov_b32", "v_xor_b32", "v_cndmask_b32 (vcc)", "v_add_f32 dpp quad_perm",
    "v_mul_f32 (VOP2)", "v_add_f32 (VOP2)", "alternating v_pk_fma_f32 / v_fma_f32", "alternating v_pk_fma_f32 / v_mov_b32",
    "v_pk_fma_f32 dependent (distance 1)", "v_pk_fma_f32 dependent (distance 2)", "v_pk_fma_f32 dependent (distance 4)",
    "v_pk_fma_f32 + s_nop 0 pairs (per pair)", "v_pk_fma_f32 (three distinct vgpr sources)", "v_mov_b64",
    "v_fma_f32 (three distinct vgpr sources)", "v_permlane32_swap", "v_cndmask_b32_e64 (sgpr-pair mask)",
    "v_cndmask_b32 (vcc written before the loop)", "v_readlane_b32", "v_writelane_b32", "v_lshl_add_u64",
    "v_pk_fma_f32 (src1 == src2 register)", "v_fma_f32 (dst, v, s, v) as x-shear scalar form"};

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, uint64_t* cyc, uint64_t* rt, float x) {
  v2f a0{x, 1.f}, a1{x, 2.f}, a2{x, 3.f}, a3{x, 4.f}, a4{x, 5.f}, a5{x, 6.f}, a6{x, 7.f}, a7{x, 8.f};
  v2f c{1.0001f, 0.0001f}, d{0.5f, 0.25f};
  uint64_t t0, t1, r0, r1;
  int s0 = 0, s1 = 0;
  uint64_t msk;
  asm volatile("s_mov_b64 %0, 0x55aa33cc" : "=s"(msk));
  asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0));
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define OPSX "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
  for (int i = 0; i < kIters; ++i) {
    if (MODE == FMA_V) {
#define T(A, B) "v_fma_f32 %" #A ", %" #A ", %8, %8\n"
      asm volatile(R16(T) : OPSX : "v"(c.x));
#undef T
    } else if (MODE == FMA_S) {
#define T(A, B) "v_fma_f32 %" #A ", %" #A ", %8, %" #A "\n"
      asm volatile(R16(T) : OPSX : "s"(c.x));
#undef T
    } else if (MODE == FMAC_V) {
#define T(A, B) "v_fmac_f32 %" #A ", %8, %9\n"
      asm volatile(R16(T) : OPSX : "v"(c.x), "v"(d.x));
#undef T
    } else if (MODE == PKFMA_V) {
#define T(A, B) "v_pk_fma_f32 %" #A ", %" #A ", %8, %8\n"
      asm volatile(R16(T) : OPS : "v"(c));
#undef T
    } else if (MODE == PKFMA_S) {
#define T(A, B) "v_pk_fma_f32 %" #A ", %" #A ", %8, %" #A "\n"
      asm volatile(R16(T) : OPS : "s"(c));
#undef T
    } else if (MODE == PKFMA_SEL) {
#define T(A, B) "v_pk_fma_f32 %" #A ", %" #B ", %8, %" #A " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n"
      asm volatile(R16(T) : OPS : "s"(c));
#undef T
    } else if (MODE == PKMUL_V) {
#define T(A, B) "v_pk_mul_f32 %" #A ", %" #A ", %8\n"
      asm volatile(R16(T) : OPS : "v"(c));
#undef T
    } else if (MODE == PKADD_V) {
#define T(A, B) "v_pk_add_f32 %" #A ", %" #A ", %8\n"
      asm volatile(R16(T) : OPS : "v"(c));
#undef T
    } else if (MODE == MOV) {
#define T(A, B) "v_mov_b32 %" #A ", %" #B "\n"
      asm volatile(R16(T) : OPSX);
#undef T
    } else if (MODE == XOR) {
#define T(A, B) "v_xor_b32 %" #A ", %" #A ", %8\n"
      asm volatile(R16(T) : OPSX : "v"(c.x));
#undef T
    } else if (MODE == CNDMASK) {
#define T(A, B) "v_cndmask_b32 %" #A ", %" #A ", %8, vcc\n"
      asm volatile(R16(T) : OPSX : "v"(c.x) : "vcc");
#undef T
    } else if (MODE == ADD_DPP) {
#define T(A, B) "v_add_f32_dpp %" #A ", %" #B ", %" #A " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      asm volatile(R16(T) : OPSX);
#undef T
    } else if (MODE == MUL_V) {
#define T(A, B) "v_mul_f32 %" #A ", %" #A ", %8\n"
      asm volatile(R16(T) : OPSX : "v"(c.x));
#undef T
    } else if (MODE == ADD_V) {
#define T(A, B) "v_add_f32 %" #A ", %" #A ", %8\n"
      asm volatile(R16(T) : OPSX : "v"(c.x));
#undef T
    } else if (MODE == MIX_PK_FMA) {
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %8\n v_fma_f32 %9, %9, %10, %10\n v_pk_fma_f32 %1, %1, %8, %8\n v_fma_f32 %11, %11, %10, %10\n"
          "v_pk_fma_f32 %2, %2, %8, %8\n v_fma_f32 %12, %12, %10, %10\n v_pk_fma_f32 %3, %3, %8, %8\n v_fma_f32 %13, %13, %10, %10\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_fma_f32 %9, %9, %10, %10\n v_pk_fma_f32 %1, %1, %8, %8\n v_fma_f32 %11, %11, %10, %10\n"
          "v_pk_fma_f32 %2, %2, %8, %8\n v_fma_f32 %12, %12, %10, %10\n v_pk_fma_f32 %3, %3, %8, %8\n v_fma_f32 %13, %13, %10, %10\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(c), "v"(a4.x), "v"(d.x), "v"(a5.x), "v"(a6.x), "v"(a7.x));
    } else if (MODE == MIX_PK_MOV) {
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %8\n v_mov_b32 %9, %10\n v_pk_fma_f32 %1, %1, %8, %8\n v_mov_b32 %10, %11\n"
          "v_pk_fma_f32 %2, %2, %8, %8\n v_mov_b32 %11, %12\n v_pk_fma_f32 %3, %3, %8, %8\n v_mov_b32 %12, %9\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_mov_b32 %9, %10\n v_pk_fma_f32 %1, %1, %8, %8\n v_mov_b32 %10, %11\n"
          "v_pk_fma_f32 %2, %2, %8, %8\n v_mov_b32 %11, %12\n v_pk_fma_f32 %3, %3, %8, %8\n v_mov_b32 %12, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(c), "v"(a4.x), "v"(a5.x), "v"(a6.x), "v"(a7.x));
    } else if (MODE == PKFMA_DEP1) {
#define T(A, B) "v_pk_fma_f32 %0, %0, %8, %8\n"
      asm volatile(R16(T) : OPS : "v"(c));
#undef T
    } else if (MODE == PKFMA_DEP2) {
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n"
          : OPS : "v"(c));
    } else if (MODE == PKFMA_DEP4) {
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
          "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
          : OPS : "v"(c));
    } else if (MODE == SNOP_PK) {
#define T(A, B) "v_pk_fma_f32 %" #A ", %" #A ", %8, %8\n s_nop 0\n"
      asm volatile(R16(T) : OPS : "v"(c));
#undef T
    } else if (MODE == PKFMA_3SRC) {
#define T(A, B) "v_pk_fma_f32 %" #A ", %" #B ", %8, %" #A "\n"
      asm volatile(R16(T) : OPS : "v"(c));
#undef T
    } else if (MODE == MOV64) {
#define T(A, B) "v_mov_b64 %" #A ", %" #B "\n"
      asm volatile(R16(T) : OPS);
#undef T
    } else if (MODE == FMA_3SRC) {
#define T(A, B) "v_fma_f32 %" #A ", %" #B ", %8, %" #A "\n"
      asm volatile(R16(T) : OPSX : "v"(c.x));
#undef T
    } else if (MODE == CND_E64) {
#define T(A, B) "v_cndmask_b32_e64 %" #A ", %" #A ", %8, %9\n"
      asm volatile(R16(T) : OPSX : "v"(c.x), "s"(msk));
#undef T
    } else if (MODE == CND_VCCSET) {
#define T(A, B) "v_cndmask_b32 %" #A ", %" #A ", %8, vcc\n"
      asm volatile("s_mov_b64 vcc, %9\n" R16(T) : OPSX : "v"(c.x), "s"(msk) : "vcc");
#undef T
    } else if (MODE == READLANE) {
      asm volatile(
          "v_readlane_b32 %0, %2, 1\n v_readlane_b32 %1, %2, 2\n v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %2, 4\n"
          "v_readlane_b32 %0, %2, 5\n v_readlane_b32 %1, %2, 6\n v_readlane_b32 %0, %2, 7\n v_readlane_b32 %1, %2, 8\n"
          "v_readlane_b32 %0, %2, 9\n v_readlane_b32 %1, %2, 10\n v_readlane_b32 %0, %2, 11\n v_readlane_b32 %1, %2, 12\n"
          "v_readlane_b32 %0, %2, 13\n v_readlane_b32 %1, %2, 14\n v_readlane_b32 %0, %2, 15\n v_readlane_b32 %1, %2, 16\n"
          : "=s"(s0), "=s"(s1) : "v"(a0.x));
    } else if (MODE == WRITELANE) {
#define T(A, B) "v_writelane_b32 %" #A ", %8, " #B "\n"
      asm volatile(R16(T) : OPSX : "s"(s0));
#undef T
    } else if (MODE == LSHL_ADD_U64) {
#define T(A, B) "v_lshl_add_u64 %" #A ", %" #B ", 2, %8\n"
      asm volatile(R16(T) : OPS : "s"(msk));
#undef T
    } else if (MODE == PKFMA_BANK) {
#define T(A, B) "v_pk_fma_f32 %" #A ", %" #B ", %" #B ", %" #A "\n"
      asm volatile(R16(T) : OPS);
#undef T
    } else if (MODE == FMA_SS) {
#define T(A, B) "v_fma_f32 %" #A ", %" #B ", %8, %" #A "\n"
      asm volatile(R16(T) : OPSX : "s"(c.x));
#undef T
    } else if (MODE == PERMLANE) {
#define T(A, B) "v_permlane32_swap_b32 %" #A ", %" #B "\n"
      asm volatile(R16(T) : OPSX);
#undef T
    }
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) { cyc[gw] = t1 - t0; rt[gw] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a1.y + a2.y + a3.y + a4.y + a5.y + a6.y + a7.y + float(s0 + s1);
}

struct Row { double cyc_per_instr, ghz, ms; };
template <int MODE>
Row run1(int wps, float* d, uint64_t* dc, uint64_t* dr) {
  const int threads = 256 * wps, blocks = 256;
  const int nw = blocks * threads / 64;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, dc, dr, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, dc, dr, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> c(nw), r(nw);
  hipMemcpy(c.data(), dc, nw * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), dr, nw * 8, hipMemcpyDeviceToHost);
  double sc = 0, sr = 0;
  for (int i = 0; i < nw; ++i) { sc += double(c[i]); sr += double(r[i]); }
  const double per = (MODE == SNOP_PK) ? 16.0 : 16.0;
  Row row;
  row.cyc_per_instr = sc / nw / (double(kIters) * per * wps);
  row.ghz = sc / sr * 0.1;  // s_memrealtime ticks at 100 MHz
  row.ms = ms;
  return row;
}
template <int MODE>
void run(float* d, uint64_t* dc, uint64_t* dr) {
  printf("%-44s", kNames[MODE]);
  for (int wps : {1, 2, 4}) {
    const Row r = run1<MODE>(wps, d, dc, dr);
    // two readings: mean wave cycles / instructions of the SIMD (optimistic when waves finish at different times),
    // and kernel time (HIP events) x clock / instructions of the SIMD (includes ~5 us of launch)
    printf(" | %d w/SIMD: %5.2f (wave) %5.2f (kernel) cyc  %.3f GHz", wps, r.cyc_per_instr,
           r.ms * 1e6 * r.ghz / (double(kIters) * 16 * wps), r.ghz);
  }
  printf("\n");
  fflush(stdout);
}
template <int... M> void run_all(float* d, uint64_t* dc, uint64_t* dr, std::integer_sequence<int, M...>) { (run<M>(d, dc, dr), ...); }

int main() {
  float* d; uint64_t *dc, *dr;
  hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&dc, 256 * 16 * 8); hipMalloc(&dr, 256 * 16 * 8);
  printf("cycles per wave-instruction per SIMD (s_memtime), sustained clock (s_memtime / s_memrealtime), kernel ms (HIP events)\n");
  run_all(d, dc, dr, std::make_integer_sequence<int, N_MODES>{});
  // the packed-fp32 rate the chip sustains: 4 flop x 64 lanes per v_pk_fma_f32
  const Row r = run1<PKFMA_S>(4, d, dc, dr);
  const double tf = 256.0 / r.cyc_per_instr * r.ghz * 1e9 * 1024 / 1e12;
  printf("sustained packed-fp32 FMA rate (sgpr operand, 4 waves/SIMD): %.1f TFLOP/s = %.3f of 157.3\n", tf, tf / 157.3);
  const Row r2 = run1<FMA_V>(4, d, dc, dr);
  const double tf2 = 128.0 / r2.cyc_per_instr * r2.ghz * 1e9 * 1024 / 1e12;
  printf("sustained scalar-fp32 FMA rate (4 waves/SIMD): %.1f TFLOP/s = %.3f of 157.3\n", tf2, tf2 / 157.3);
  return 0;
}
