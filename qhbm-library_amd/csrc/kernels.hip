// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the expectation engine.
//
// Replaces the body of tfq.layers.Expectation (forward and adjoint backward),
// called from /root/reference/qhbmlib/inference/qnn.py:134-138; the reference
// runs it on CPU threads over serialized circuits.
//
// Execution model (see program.h): one workgroup = one LDS-resident tile of
// 2^K amplitudes of one statevector; wave64; all control flow is driven by a
// wave-uniform program so branches are scalar.  Amplitudes are complex64
// (float2, re/im interleaved) in HBM and in LDS.
//
// LDS layout of a tile: amplitude with local index l lives at slot
//     swz(l) = l ^ ((l >> 5) & 31)
// XOR-ing index bits 5..9 into bits 0..4 makes 8-byte accesses conflict-free
// both when a half-wave walks the low bits (stride 1) and when it walks bits
// 5..9 (stride 32) -- the two patterns the rounds produce.  swz is linear over
// XOR, so swz(thread_part | reg_part) = swz(thread_part) ^ swz(reg_part): one
// v_xor per LDS address.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstring>

#include <utility>

#include "../../include/qhbm_engine.h"
#include "device_common.h"
#include "kernels.h"
#include "program.h"

namespace qhbm {

namespace {


__device__ __forceinline__ uint32_t swz(uint32_t l) { return l ^ ((l >> 5) & 31u); }

// Waves per SIMD the register allocator must leave room for: as many workgroups per CU
// as the LDS footprint admits (160 KiB per CU), capped at 4 waves per SIMD.
constexpr int wg_per_cu(int lds_bytes) { return (160 * 1024) / lds_bytes < 1 ? 1 : (160 * 1024) / lds_bytes; }
constexpr int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
constexpr int fwd_lds(int K) { return (8 << K) + 8 * kMaxOps; }
// adjoint, two-tile layout: (psi, lambda) tiles + one gradient cell per slot and wave
constexpr int adj_lds(int K) { return (16 << K) + 4 * kMaxSlotsPerPass * ((1 << (K - 4)) / 64 < 1 ? 1 : (1 << (K - 4)) / 64); }
// adjoint, exchange layout: ONE tile-sized exchange buffer (the tile pair lives in registers)
constexpr int adjx_lds(int K) { return (8 << K) + 4 * kMaxSlotsPerPass * ((1 << (K - 4)) / 64 < 1 ? 1 : (1 << (K - 4)) / 64); }
constexpr int fwd_min_waves(int K, int R) {
  return clampi(wg_per_cu(fwd_lds(K)) * (1 << (K - R)) / 256, 1, 4);
}
constexpr int adj_min_waves(int K) { return clampi(wg_per_cu(adj_lds(K)) * (1 << (K - 4)) / 256, 1, 4); }
constexpr int adjx_min_waves(int K) { return clampi(wg_per_cu(adjx_lds(K)) * (1 << (K - 4)) / 256, 1, 4); }

// Static iteration over register bits / register-bit pairs WITHOUT lambdas (a lambda
// capturing the register arrays by reference pins them to scratch memory).
#define QHBM_FOR_RB(R_, ...)                              \
  {                                                       \
    { constexpr int J = 0; __VA_ARGS__ }                  \
    { constexpr int J = 1; __VA_ARGS__ }                  \
    { constexpr int J = 2; __VA_ARGS__ }                  \
    { constexpr int J = 3; __VA_ARGS__ }                  \
    if constexpr ((R_) > 4) { constexpr int J = (R_) > 4 ? 4 : 0; __VA_ARGS__ } \
  }
#define QHBM_PAIR(JA_, JB_, ...) { constexpr int JA = JA_; constexpr int JB = JB_; __VA_ARGS__ }
#define QHBM_FOR_PAIR(R_, ...)                            \
  {                                                       \
    QHBM_PAIR(0, 1, __VA_ARGS__) QHBM_PAIR(0, 2, __VA_ARGS__) QHBM_PAIR(1, 2, __VA_ARGS__) \
    QHBM_PAIR(0, 3, __VA_ARGS__) QHBM_PAIR(1, 3, __VA_ARGS__) QHBM_PAIR(2, 3, __VA_ARGS__) \
    if constexpr ((R_) > 4) {                             \
      QHBM_PAIR(0, ((R_) > 4 ? 4 : 1), __VA_ARGS__) QHBM_PAIR(1, ((R_) > 4 ? 4 : 2), __VA_ARGS__) \
      QHBM_PAIR(2, ((R_) > 4 ? 4 : 3), __VA_ARGS__) QHBM_PAIR(((R_) > 4 ? 3 : 0), ((R_) > 4 ? 4 : 1), __VA_ARGS__) \
    }                                                     \
  }

// register index with a zero inserted at bit RB
template <int RB> constexpr int ins0(int p) { return ((p >> RB) << (RB + 1)) | (p & ((1 << RB) - 1)); }

// ---- in-register gate kernels --------------------------------------------------
// One amplitude = one 64-bit VGPR pair (re, im).  The hot updates are in-place
// VOP3P packed-fp32 sequences: op_sel picks which half of a source feeds the
// low/high result, neg_lo/neg_hi flip a sign, so a complex multiply-add is one
// v_pk_mul + one v_pk_fma and nothing is ever re-packed.  cs = (c, s) is a
// wave-uniform SGPR pair, or a VGPR pair for thread-predicated phases.

// Y**t on a pair: (a0, a1) <- (c a0 - s a1, s a0 + c a1).  Rotated products go to temporaries
// first, then each amplitude is updated in place -- four packed ops per pair, no register copy.
__device__ __forceinline__ void y_pair(v2f& a0, v2f& a1, v2f cs) {
  v2f t0, t1;
  asm("v_pk_mul_f32 %[t0], %[a0], %[cs] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[t1], %[a1], %[cs] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_fma_f32 %[a0], %[a0], %[cs], %[t1] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[a1], %[a1], %[cs], %[t0] op_sel_hi:[1,0,1]"
      : [a0] "+v"(a0), [a1] "+v"(a1), [t0] "=&v"(t0), [t1] "=&v"(t1)
      : [cs] "s"(cs));
}

// a <- (c + i s) a = c a + s (-a.im, a.re)
__device__ __forceinline__ void phase_s(v2f& a, v2f cs) {
  v2f t;
  asm("v_pk_mul_f32 %[t], %[a], %[cs] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %[a], %[a], %[cs], %[t] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : [a] "+v"(a), [t] "=&v"(t)
      : [cs] "s"(cs));
}
__device__ __forceinline__ v2f conj_cs(v2f cs) { return v2f{cs.x, -cs.y}; }

// All loops over the register file are integer_sequence folds: every index is a
// constant when the IR is generated, so the array is scalarised into independent
// VGPR pairs (a run-time loop index, even in a fully unrolled loop, makes the
// compiler keep the file as ONE 1024-bit tuple and copy it on every update).
template <int N> using iseq = std::make_integer_sequence<int, N>;

// X**t core c*I - i*s*X (theta = pi t / 2 reduced to [-pi/2, pi/2], prep_coefs_kernel) as THREE SHEARS,
//   [[c, -is], [-is, c]] = [[1, u], [0, 1]] [[1, 0], [v, 1]] [[1, u], [0, 1]],  u = -i tan(theta/2), v = -i sin(theta),
// each one in-place packed FMA per pair (-i w (x + i y) = (w y, -w x)): 1.5 packed ops per amplitude
// instead of the 2 of the direct form, no temporaries, determinant exactly 1.  ts = (tan(theta/2),
// sin(theta)) is a wave-uniform SGPR pair; U^dagger is the same with ts negated.  Four pairs per asm
// statement: the compiler pads every inline-asm block with an s_nop (it cannot see the hazards
// inside), so fewer, longer blocks issue fewer of them, and dependent FMAs sit four apart.
__device__ __forceinline__ void x_pair4(v2f& a0, v2f& a1, v2f& b0, v2f& b1, v2f& c0, v2f& c1, v2f& d0, v2f& d1, v2f ts) {
#define QHBM_SH_T(D_, S_) "v_pk_fma_f32 %[" #D_ "], %[" #S_ "], %[ts], %[" #D_ "] op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]\n\t"
#define QHBM_SH_S(D_, S_) "v_pk_fma_f32 %[" #D_ "], %[" #S_ "], %[ts], %[" #D_ "] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
  asm(QHBM_SH_T(a0, a1) QHBM_SH_T(b0, b1) QHBM_SH_T(c0, c1) QHBM_SH_T(d0, d1)
      QHBM_SH_S(a1, a0) QHBM_SH_S(b1, b0) QHBM_SH_S(c1, c0) QHBM_SH_S(d1, d0)
      QHBM_SH_T(a0, a1) QHBM_SH_T(b0, b1) QHBM_SH_T(c0, c1)
      "v_pk_fma_f32 %[d0], %[d1], %[ts], %[d0] op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]"
      : [a0] "+v"(a0), [a1] "+v"(a1), [b0] "+v"(b0), [b1] "+v"(b1), [c0] "+v"(c0), [c1] "+v"(c1),
        [d0] "+v"(d0), [d1] "+v"(d1)
      : [ts] "s"(ts));
#undef QHBM_SH_T
#undef QHBM_SH_S
}
template <int R, int RB, int... Q>
__device__ __forceinline__ void apply_x_(v2f (&a)[1 << R], v2f cs, std::integer_sequence<int, Q...>) {
  static_assert(R == 4, "x_pair4 covers the eight pairs of a 16-amplitude register file in two calls");
  (x_pair4(a[ins0<RB>(4 * Q)], a[ins0<RB>(4 * Q) | (1 << RB)], a[ins0<RB>(4 * Q + 1)], a[ins0<RB>(4 * Q + 1) | (1 << RB)],
           a[ins0<RB>(4 * Q + 2)], a[ins0<RB>(4 * Q + 2) | (1 << RB)], a[ins0<RB>(4 * Q + 3)],
           a[ins0<RB>(4 * Q + 3) | (1 << RB)], cs), ...);
}
// c*I - i*s*X  on register bit RB  (cs = (tan(theta/2), sin(theta)), see x_pair4)
template <int R, int RB>
__device__ __forceinline__ void apply_x(v2f (&a)[1 << R], v2f cs) { apply_x_<R, RB>(a, cs, iseq<(1 << (R - 3))>{}); }

template <int R, int RB, int... P>
__device__ __forceinline__ void apply_y_(v2f (&a)[1 << R], v2f cs, std::integer_sequence<int, P...>) {
  (y_pair(a[ins0<RB>(P)], a[ins0<RB>(P) | (1 << RB)], cs), ...);
}
// c*I - i*s*Y = [[c, -s], [s, c]]
template <int R, int RB>
__device__ __forceinline__ void apply_y(v2f (&a)[1 << R], v2f cs) { apply_y_<R, RB>(a, cs, iseq<(1 << (R - 1))>{}); }

// dense 2x2 in place: u00..u11 are wave-uniform complex numbers (SGPR pairs).
// n0 = u00 a0 + u01 a1, n1 = u10 a0 + u11 a1;  u*a = u.re * a + u.im * (-a.im, a.re)
__device__ __forceinline__ void mat1_pair(v2f& a0, v2f& a1, v2f u00, v2f u01, v2f u10, v2f u11) {
  v2f t0, t1;
  asm("v_pk_mul_f32 %[t0], %[a0], %[u00] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %[t0], %[a0], %[u00], %[t0] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
      "v_pk_fma_f32 %[t0], %[a1], %[u01], %[t0] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[t0], %[a1], %[u01], %[t0] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
      "v_pk_mul_f32 %[t1], %[a0], %[u10] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %[t1], %[a0], %[u10], %[t1] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
      "v_pk_fma_f32 %[t1], %[a1], %[u11], %[t1] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[t1], %[a1], %[u11], %[t1] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
      "v_mov_b64 %[a0], %[t0]\n\t"
      "v_mov_b64 %[a1], %[t1]"
      : [a0] "+v"(a0), [a1] "+v"(a1), [t0] "=&v"(t0), [t1] "=&v"(t1)
      : [u00] "s"(u00), [u01] "s"(u01), [u10] "s"(u10), [u11] "s"(u11));
}
template <int R, int RB, int... P>
__device__ __forceinline__ void apply_mat1_(v2f (&a)[1 << R], v2f u00, v2f u01, v2f u10, v2f u11,
                                            std::integer_sequence<int, P...>) {
  (mat1_pair(a[ins0<RB>(P)], a[ins0<RB>(P) | (1 << RB)], u00, u01, u10, u11), ...);
}
// dense 2x2 (rare path), u = row-major wave-uniform complex entries
template <int R, int RB>
__device__ __forceinline__ void apply_mat1(v2f (&a)[1 << R], v2f u00, v2f u01, v2f u10, v2f u11) {
  apply_mat1_<R, RB>(a, u00, u01, u10, u11, iseq<(1 << (R - 1))>{});
}

template <int R, int RB, int... P>
__device__ __forceinline__ void apply_ph1_(v2f (&a)[1 << R], v2f cs, std::integer_sequence<int, P...>) {
  (phase_s(a[ins0<RB>(P) | (1 << RB)], cs), ...);
}
// multiply by (c + i s) the amplitudes whose register bit RB is 1 (uniform coefficient)
template <int R, int RB>
__device__ __forceinline__ void apply_ph1(v2f (&a)[1 << R], v2f cs) { apply_ph1_<R, RB>(a, cs, iseq<(1 << (R - 1))>{}); }

// a_k <- (c + i s) a_k for eight amplitudes in one asm statement (see x_pair4).
__device__ __forceinline__ void phase_v8(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f& a4, v2f& a5, v2f& a6, v2f& a7, v2f cs) {
  v2f t0, t1, t2, t3, t4, t5, t6, t7;
#define QHBM_PH_MUL(K_) "v_pk_mul_f32 %[t" #K_ "], %[a" #K_ "], %[cs] op_sel_hi:[1,0]\n\t"
#define QHBM_PH_FMA(K_) "v_pk_fma_f32 %[a" #K_ "], %[a" #K_ "], %[cs], %[t" #K_ "] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
  asm(QHBM_PH_MUL(0) QHBM_PH_MUL(1) QHBM_PH_MUL(2) QHBM_PH_MUL(3) QHBM_PH_MUL(4) QHBM_PH_MUL(5) QHBM_PH_MUL(6) QHBM_PH_MUL(7)
      QHBM_PH_FMA(0) QHBM_PH_FMA(1) QHBM_PH_FMA(2) QHBM_PH_FMA(3) QHBM_PH_FMA(4) QHBM_PH_FMA(5) QHBM_PH_FMA(6) "v_pk_fma_f32 %[a7], %[a7], %[cs], %[t7] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),
        [a7] "+v"(a7), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
        [t6] "=&v"(t6), [t7] "=&v"(t7)
      : [cs] "v"(cs));
#undef QHBM_PH_MUL
#undef QHBM_PH_FMA
}
// ... with a wave-uniform coefficient (SGPR pair): the boundary phases run it under the predicate's EXEC mask
// instead of selecting a per-lane coefficient (two v_mov + two v_cndmask per micro-op, VCC ones in the forward
// kernels: 23 cycles each on gfx950, scripts/experiments/micro/valu_cycles.hip).
__device__ __forceinline__ void phase_s8(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f& a4, v2f& a5, v2f& a6, v2f& a7, v2f cs) {
  v2f t0, t1, t2, t3, t4, t5, t6, t7;
#define QHBM_PH_MUL(K_) "v_pk_mul_f32 %[t" #K_ "], %[a" #K_ "], %[cs] op_sel_hi:[1,0]\n\t"
#define QHBM_PH_FMA(K_) "v_pk_fma_f32 %[a" #K_ "], %[a" #K_ "], %[cs], %[t" #K_ "] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
  asm(QHBM_PH_MUL(0) QHBM_PH_MUL(1) QHBM_PH_MUL(2) QHBM_PH_MUL(3) QHBM_PH_MUL(4) QHBM_PH_MUL(5) QHBM_PH_MUL(6) QHBM_PH_MUL(7)
      QHBM_PH_FMA(0) QHBM_PH_FMA(1) QHBM_PH_FMA(2) QHBM_PH_FMA(3) QHBM_PH_FMA(4) QHBM_PH_FMA(5) QHBM_PH_FMA(6) "v_pk_fma_f32 %[a7], %[a7], %[cs], %[t7] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),
        [a7] "+v"(a7), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
        [t6] "=&v"(t6), [t7] "=&v"(t7)
      : [cs] "s"(cs));
#undef QHBM_PH_MUL
#undef QHBM_PH_FMA
}
template <int R, int RB>
__device__ __forceinline__ void apply_ph1_s8(v2f (&a)[1 << R], v2f cs) {
  static_assert(R == 4, "eight amplitudes with register bit RB set");
  constexpr int B = 1 << RB;
  phase_s8(a[ins0<RB>(0) | B], a[ins0<RB>(1) | B], a[ins0<RB>(2) | B], a[ins0<RB>(3) | B], a[ins0<RB>(4) | B],
           a[ins0<RB>(5) | B], a[ins0<RB>(6) | B], a[ins0<RB>(7) | B], cs);
}
template <int R, int RB>
__device__ __forceinline__ void apply_ph1_v_(v2f (&a)[1 << R], v2f cs, std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>) {
  constexpr int B = 1 << RB;
  phase_v8(a[ins0<RB>(0) | B], a[ins0<RB>(1) | B], a[ins0<RB>(2) | B], a[ins0<RB>(3) | B], a[ins0<RB>(4) | B],
           a[ins0<RB>(5) | B], a[ins0<RB>(6) | B], a[ins0<RB>(7) | B], cs);
}
// same with a per-thread coefficient
template <int R, int RB>
__device__ __forceinline__ void apply_ph1_v(v2f (&a)[1 << R], v2f cs) { apply_ph1_v_<R, RB>(a, cs, iseq<(1 << (R - 1))>{}); }

template <int RA, int RB> constexpr int ins11(int p) { return ins0<RB>(ins0<RA>(p)) | (1 << RA) | (1 << RB); }
template <int R, int RA, int RB, int... P>
__device__ __forceinline__ void apply_ph2_(v2f (&a)[1 << R], v2f cs, std::integer_sequence<int, P...>) {
  (phase_s(a[ins11<RA, RB>(P)], cs), ...);
}
// ... whose register bits RA < RB are both 1
template <int R, int RA, int RB>
__device__ __forceinline__ void apply_ph2(v2f (&a)[1 << R], v2f cs) { apply_ph2_<R, RA, RB>(a, cs, iseq<(1 << (R - 2))>{}); }

__device__ __forceinline__ float im_conj(v2f l, v2f p) { return l.x * p.y - l.y * p.x; }  // Im(conj(l) p)
// acc += (l.re * p.im, l.im * p.re): Im(conj(l) p) = acc.x - acc.y summed later -- ONE packed FMA
// per amplitude instead of mul + fma + add.
__device__ __forceinline__ void acc_im(v2f& acc, v2f l, v2f p) {
  asm("v_pk_fma_f32 %[acc], %[l], %[p], %[acc] op_sel:[0,1,0] op_sel_hi:[1,0,1]"
      : [acc] "+v"(acc) : [l] "v"(l), [p] "v"(p));
}
// acc += (l.re * p.re, l.im * p.im): Re(conj(l) p) = acc.x + acc.y
__device__ __forceinline__ void acc_re(v2f& acc, v2f l, v2f p) {
  asm("v_pk_fma_f32 %[acc], %[l], %[p], %[acc]" : [acc] "+v"(acc) : [l] "v"(l), [p] "v"(p));
}

// sum over selected registers of Im(conj(lam) psi); two packed accumulators break the chain
// Eight accumulations into two partial sums in one asm statement (see x_pair4).
// The same with the two partial sums STARTED here (products 0 and 1 are multiplies): no zeroed accumulators
// (two v_mov_b64 per sum: 4.1 cycles each on gfx950, scripts/experiments/micro/valu_cycles.hip).
__device__ __forceinline__ void acc_im8_first(v2f& a0, v2f& a1, v2f l0, v2f p0, v2f l1, v2f p1, v2f l2, v2f p2, v2f l3, v2f p3,
                                              v2f l4, v2f p4, v2f l5, v2f p5, v2f l6, v2f p6, v2f l7, v2f p7) {
#define QHBM_ACC(A_, K_) "v_pk_fma_f32 %[" #A_ "], %[l" #K_ "], %[p" #K_ "], %[" #A_ "] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
  asm("v_pk_mul_f32 %[a0], %[l0], %[p0] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[a1], %[l1], %[p1] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      QHBM_ACC(a0, 2) QHBM_ACC(a1, 3) QHBM_ACC(a0, 4) QHBM_ACC(a1, 5) QHBM_ACC(a0, 6)
      "v_pk_fma_f32 %[a1], %[l7], %[p7], %[a1] op_sel:[0,1,0] op_sel_hi:[1,0,1]"
      : [a0] "=&v"(a0), [a1] "=&v"(a1)
      : [l0] "v"(l0), [p0] "v"(p0), [l1] "v"(l1), [p1] "v"(p1), [l2] "v"(l2), [p2] "v"(p2), [l3] "v"(l3), [p3] "v"(p3),
        [l4] "v"(l4), [p4] "v"(p4), [l5] "v"(l5), [p5] "v"(p5), [l6] "v"(l6), [p6] "v"(p6), [l7] "v"(l7), [p7] "v"(p7));
#undef QHBM_ACC
}
__device__ __forceinline__ void acc_im8(v2f& a0, v2f& a1, v2f l0, v2f p0, v2f l1, v2f p1, v2f l2, v2f p2, v2f l3, v2f p3,
                                        v2f l4, v2f p4, v2f l5, v2f p5, v2f l6, v2f p6, v2f l7, v2f p7) {
#define QHBM_ACC(A_, K_) "v_pk_fma_f32 %[" #A_ "], %[l" #K_ "], %[p" #K_ "], %[" #A_ "] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
  asm(QHBM_ACC(a0, 0) QHBM_ACC(a1, 1) QHBM_ACC(a0, 2) QHBM_ACC(a1, 3) QHBM_ACC(a0, 4) QHBM_ACC(a1, 5) QHBM_ACC(a0, 6)
      "v_pk_fma_f32 %[a1], %[l7], %[p7], %[a1] op_sel:[0,1,0] op_sel_hi:[1,0,1]"
      : [a0] "+v"(a0), [a1] "+v"(a1)
      : [l0] "v"(l0), [p0] "v"(p0), [l1] "v"(l1), [p1] "v"(p1), [l2] "v"(l2), [p2] "v"(p2), [l3] "v"(l3), [p3] "v"(p3),
        [l4] "v"(l4), [p4] "v"(p4), [l5] "v"(l5), [p5] "v"(p5), [l6] "v"(l6), [p6] "v"(p6), [l7] "v"(l7), [p7] "v"(p7));
#undef QHBM_ACC
}
template <int R, int RB>
__device__ __forceinline__ float sum_w1_(const v2f (&p)[1 << R], const v2f (&l)[1 << R], std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>) {
  constexpr int B = 1 << RB;
  v2f a0, a1;
  acc_im8_first(a0, a1, l[ins0<RB>(0) | B], p[ins0<RB>(0) | B], l[ins0<RB>(1) | B], p[ins0<RB>(1) | B], l[ins0<RB>(2) | B],
          p[ins0<RB>(2) | B], l[ins0<RB>(3) | B], p[ins0<RB>(3) | B], l[ins0<RB>(4) | B], p[ins0<RB>(4) | B],
          l[ins0<RB>(5) | B], p[ins0<RB>(5) | B], l[ins0<RB>(6) | B], p[ins0<RB>(6) | B], l[ins0<RB>(7) | B],
          p[ins0<RB>(7) | B]);
  return (a0.x + a1.x) - (a0.y + a1.y);
}
template <int R, int RB>
__device__ __forceinline__ float sum_w1(const v2f (&p)[1 << R], const v2f (&l)[1 << R]) {
  return sum_w1_<R, RB>(p, l, iseq<(1 << (R - 1))>{});
}
template <int R, int RA, int RB, int... P>
__device__ __forceinline__ float sum_w2_(const v2f (&p)[1 << R], const v2f (&l)[1 << R], std::integer_sequence<int, P...>) {
  static_assert(sizeof...(P) == 4, "four amplitudes with both register bits set");
  v2f a0, a1;
  asm("v_pk_mul_f32 %[a0], %[l0], %[p0] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %[a1], %[l1], %[p1] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %[a0], %[l2], %[p2], %[a0] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[a1], %[l3], %[p3], %[a1] op_sel:[0,1,0] op_sel_hi:[1,0,1]"
      : [a0] "=&v"(a0), [a1] "=&v"(a1)
      : [l0] "v"(l[ins11<RA, RB>(0)]), [p0] "v"(p[ins11<RA, RB>(0)]), [l1] "v"(l[ins11<RA, RB>(1)]), [p1] "v"(p[ins11<RA, RB>(1)]),
        [l2] "v"(l[ins11<RA, RB>(2)]), [p2] "v"(p[ins11<RA, RB>(2)]), [l3] "v"(l[ins11<RA, RB>(3)]), [p3] "v"(p[ins11<RA, RB>(3)]));
  return (a0.x + a1.x) - (a0.y + a1.y);
}
template <int R, int RA, int RB>
__device__ __forceinline__ float sum_w2(const v2f (&p)[1 << R], const v2f (&l)[1 << R]) {
  return sum_w2_<R, RA, RB>(p, l, iseq<(1 << (R - 2))>{});
}

// Im <lam| G |psi> restricted to this thread's registers, for the generators
// of the fast-path gates.  X: pairs swap.  Y: (Y psi)_0 = -i psi_1, (Y psi)_1 = i psi_0.
template <int R, int RB, int... M>
__device__ __forceinline__ float im_lam_x_psi_(const v2f (&p)[1 << R], const v2f (&l)[1 << R], std::integer_sequence<int, M...>) {
  static_assert(sizeof...(M) == 16, "two statements of eight");
  constexpr int B = 1 << RB;
  v2f a0, a1;  // Im(conj(lam_m) * psi_{m ^ bit}) over all m
  acc_im8_first(a0, a1, l[0], p[0 ^ B], l[1], p[1 ^ B], l[2], p[2 ^ B], l[3], p[3 ^ B], l[4], p[4 ^ B], l[5], p[5 ^ B], l[6],
                p[6 ^ B], l[7], p[7 ^ B]);
  acc_im8(a0, a1, l[8], p[8 ^ B], l[9], p[9 ^ B], l[10], p[10 ^ B], l[11], p[11 ^ B], l[12], p[12 ^ B], l[13], p[13 ^ B],
          l[14], p[14 ^ B], l[15], p[15 ^ B]);
  return (a0.x + a1.x) - (a0.y + a1.y);
}
template <int R, int RB>
__device__ __forceinline__ float im_lam_x_psi(const v2f (&p)[1 << R], const v2f (&l)[1 << R]) {
  return im_lam_x_psi_<R, RB>(p, l, iseq<(1 << R)>{});
}
template <int R, int RB, int... P>
__device__ __forceinline__ float im_lam_y_psi_(const v2f (&p)[1 << R], const v2f (&l)[1 << R], std::integer_sequence<int, P...>) {
  // Im(conj(l0)*(-i p1)) = -Re(conj(l0) p1);  Im(conj(l1)*(i p0)) = Re(conj(l1) p0)
  v2f pos = v2f{0.f, 0.f}, neg = v2f{0.f, 0.f};
  ((acc_re(pos, l[ins0<RB>(P) | (1 << RB)], p[ins0<RB>(P)]), acc_re(neg, l[ins0<RB>(P)], p[ins0<RB>(P) | (1 << RB)])), ...);
  return (pos.x + pos.y) - (neg.x + neg.y);
}
template <int R, int RB>
__device__ __forceinline__ float im_lam_y_psi(const v2f (&p)[1 << R], const v2f (&l)[1 << R]) {
  return im_lam_y_psi_<R, RB>(p, l, iseq<(1 << (R - 1))>{});
}
// Im sum_ij conj(lam_i) g_ij psi_j over one pair, g wave-uniform row-major complex
struct Gen2 { v2f g00, g01, g10, g11; };
__device__ __forceinline__ v2f cmul(v2f g, v2f p) { return v2f{g.x * p.x - g.y * p.y, g.x * p.y + g.y * p.x}; }
__device__ __forceinline__ float g1_pair(v2f p0, v2f p1, v2f l0, v2f l1, const Gen2& g) {
  const v2f s0 = cmul(g.g00, p0) + cmul(g.g01, p1);
  const v2f s1 = cmul(g.g10, p0) + cmul(g.g11, p1);
  return im_conj(l0, s0) + im_conj(l1, s1);
}
template <int R, int RB, int... P>
__device__ __forceinline__ float im_lam_g1_psi_(const v2f (&p)[1 << R], const v2f (&l)[1 << R], const Gen2& g, std::integer_sequence<int, P...>) {
  return (g1_pair(p[ins0<RB>(P)], p[ins0<RB>(P) | (1 << RB)], l[ins0<RB>(P)], l[ins0<RB>(P) | (1 << RB)], g) + ...);
}
template <int R, int RB>
__device__ __forceinline__ float im_lam_g1_psi(const v2f (&p)[1 << R], const v2f (&l)[1 << R], const Gen2& g) {
  return im_lam_g1_psi_<R, RB>(p, l, g, iseq<(1 << (R - 1))>{});
}

// ---- shared pieces of the pass kernels ----------------------------------------
struct TileCtx {
  uint32_t tile_base;  // nonlocal bits of this tile, in index space
  uint32_t cmask;
  uint32_t c;
  const uint32_t* spread;
  uint32_t shift;  // ~0u: look the high part up in `spread`
  uint32_t ro[8];  // tile_offset(I << (K - 3)), I = 0..7: the rows of a thread's eight float4 (PassArgs::row_off)
};

// Local index -> global index WITHOUT the tile's own bits.  Both forms are bitwise (OR-decomposable):
// f(a | b) = f(a) | f(b) for disjoint a, b.  A thread's eight float4 sit at l = 2 * tid + (I << (K - 3)):
// f(2 * tid) is computed once per thread, f(I << (K - 3)) is the same for every workgroup of the pass (the host
// computes it: PassArgs::row_off) and goes into the base pointer, so the eight accesses of a tile share one
// 32-bit VGPR offset and cost no VALU address arithmetic (global_load ... v_off, s[base]).
// (local bits above c contiguous from bit `shift`: no table lookup in front of the global access)
__device__ __forceinline__ uint32_t tile_offset(const TileCtx& t, uint32_t l) {
  return (l & t.cmask) | (t.shift != 0xffffffffu ? ((l >> t.c) << t.shift) : t.spread[l >> t.c]);
}
__device__ __forceinline__ uint32_t global_index(const TileCtx& t, uint32_t l) { return t.tile_base | tile_offset(t, l); }

// (ROWS: adjoint tail passes may have no index bit 0 among their local bits -- the two amplitudes of a float4 are
// then 8 bytes each.  ROWS16: never; ROWS8: always (the exchange kernel is instantiated for both: a run-time test
// lets the compiler merge the two paths into THREE accesses per row); ROWS_TEST: tested at run time)
enum TileRows : int { ROWS16 = 0, ROWS_TEST = 1, ROWS8 = 2 };
// A thread's offsets inside its rows (a table lookup when the tile's high local bits are scattered): computed
// ONCE for the tiles a workgroup loads -- per call, the lookup of the second tile waited (vmcnt is in order) for
// the first tile's eight loads to return before its own could be issued.
struct ThreadOff {
  uint32_t g0, g1;  // of local index 2 tid, and of 2 tid + 1 when that is not the next address (ROWS8)
};
template <int ROWS = ROWS16>
__device__ __forceinline__ ThreadOff thread_offsets(const TileCtx& t, int tid) {
  ThreadOff o;
  o.g0 = tile_offset(t, 2u * uint32_t(tid));
  o.g1 = (ROWS == ROWS8 || (ROWS == ROWS_TEST && t.c == 0)) ? (o.g0 | tile_offset(t, 1u)) : o.g0 + 1u;
  return o;
}
template <int K, int NT, int ROWS = ROWS16>
__device__ __forceinline__ void store_tile(const float2* __restrict__ tile, float2* __restrict__ st,
                                           const TileCtx& t, const ThreadOff& o, int tid) {
  static_assert((1 << (K - 1)) / NT == 8, "a thread owns eight float4 of its tile");
  const uint32_t g0 = o.g0;
  const uint32_t s0 = swz(2u * uint32_t(tid));
  if (ROWS == ROWS8 || (ROWS == ROWS_TEST && t.c == 0)) {  // index bit 0 is not local (tail passes of the adjoint sweep): two 8-byte stores
    const uint32_t g1 = o.g1;
#define QHBM_ST(I)                                                                                     \
  {                                                                                                    \
    float2* sb = st + (t.tile_base | t.ro[I]);                                                         \
    const uint32_t s = s0 ^ swz(uint32_t(I) << (K - 3));                                               \
    sb[g0] = tile[s];                                                                                  \
    sb[g1] = tile[s ^ 1u];                                                                             \
  }
    QHBM_ST(0) QHBM_ST(1) QHBM_ST(2) QHBM_ST(3) QHBM_ST(4) QHBM_ST(5) QHBM_ST(6) QHBM_ST(7)
#undef QHBM_ST
    return;
  }
#define QHBM_ST(I)                                                                                     \
  {                                                                                                    \
    float2* sb = st + (t.tile_base | t.ro[I]);                                                         \
    const uint32_t s = s0 ^ swz(uint32_t(I) << (K - 3));                                               \
    const float2 a = tile[s], b = tile[s ^ 1u];                                                        \
    *reinterpret_cast<float4*>(sb + g0) = make_float4(a.x, a.y, b.x, b.y);                             \
  }
  QHBM_ST(0) QHBM_ST(1) QHBM_ST(2) QHBM_ST(3) QHBM_ST(4) QHBM_ST(5) QHBM_ST(6) QHBM_ST(7)
#undef QHBM_ST
}

// Eight float4 of a thread's share of a tile, as named registers (a loop-carried array may end
// up in scratch).  Loading a tile pair through two of these puts all 16 global loads of a thread
// in flight before the first LDS write.
struct TileRegs {
  float4 p0, p1, p2, p3, p4, p5, p6, p7;
};
template <int K, int NT, int ROWS = ROWS16>
__device__ __forceinline__ void prefetch_tile(TileRegs& r, const float2* __restrict__ st, const TileCtx& t, const ThreadOff& o) {
  static_assert((1 << (K - 1)) / NT == 8, "a thread owns eight float4 of its tile");
  const uint32_t g0 = o.g0;
  if (ROWS == ROWS8 || (ROWS == ROWS_TEST && t.c == 0)) {  // index bit 0 is not local: the two amplitudes of a float4 are 8-byte loads
    const uint32_t g1 = o.g1;
#define QHBM_PF(I)                                                                                       \
  {                                                                                                      \
    const float2* sb = st + (t.tile_base | t.ro[I]);                                                     \
    const float2 a = sb[g0], b = sb[g1];                                                                 \
    r.p##I = make_float4(a.x, a.y, b.x, b.y);                                                            \
  }
    QHBM_PF(0) QHBM_PF(1) QHBM_PF(2) QHBM_PF(3) QHBM_PF(4) QHBM_PF(5) QHBM_PF(6) QHBM_PF(7)
#undef QHBM_PF
    return;
  }
#define QHBM_PF(I)                                                                                       \
  {                                                                                                      \
    const float2* sb = st + (t.tile_base | t.ro[I]);                                                     \
    r.p##I = *reinterpret_cast<const float4*>(sb + g0);                                                  \
  }
  QHBM_PF(0) QHBM_PF(1) QHBM_PF(2) QHBM_PF(3) QHBM_PF(4) QHBM_PF(5) QHBM_PF(6) QHBM_PF(7)
#undef QHBM_PF
}
template <int K, int NT>
__device__ __forceinline__ void commit_tile(float2* __restrict__ tile, const TileRegs& r, int tid) {
  const uint32_t s0 = swz(2u * uint32_t(tid));
#define QHBM_CM(I)                                                    \
  {                                                                   \
    const uint32_t sl = s0 ^ swz(uint32_t(I) << (K - 3));             \
    tile[sl] = make_float2(r.p##I.x, r.p##I.y);                       \
    tile[sl ^ 1u] = make_float2(r.p##I.z, r.p##I.w);                  \
  }
  QHBM_CM(0) QHBM_CM(1) QHBM_CM(2) QHBM_CM(3) QHBM_CM(4) QHBM_CM(5) QHBM_CM(6) QHBM_CM(7)
#undef QHBM_CM
}

// ---- what a workgroup derives from its block index, the pass and the input bitstring ------------------
// PassArgs' per-bit tables are bytes in the kernel-argument segment: a loop over one compiles to a vector load PER
// ITERATION, each waited for -- written as loops, the adjoint kernel ran n_user + n + 2 n_nonlocal + K dependent
// memory round trips (38 at config 3) before it could issue its first tile load.  Here every table is read one
// element per LANE (all loads in flight together, one wait) and the bit permutations are ballots.
__device__ __forceinline__ uint32_t ballot32(bool p) { return uint32_t(__builtin_amdgcn_ballot_w64(p)); }

// The input bitstring as an index in the layout this pass loads: bit n_user-1-q of the logical index = row[q],
// logical bit b sits on physical position phys_of[b] (identity unless the adjoint plan relabels).
__device__ __forceinline__ uint32_t input_index(const PassArgs& a, const int8_t* __restrict__ row, int n_user, int lane) {
  const uint32_t L = uint32_t(lane);
  const uint32_t src = a.log_of[L & 31u];
  const int8_t r = row[min(L, uint32_t(n_user) - 1u)];  // (no branch around the load: both tables in flight together)
  const uint32_t logical = __brev(ballot32((L < uint32_t(n_user)) & (r != 0))) >> (32u - uint32_t(n_user));
  return ballot32((L < a.n) & (((logical >> src) & 1u) != 0u));
}
// ... its bits on the K local positions of the tile.
template <int K>
__device__ __forceinline__ uint32_t local_bits(const PassArgs& a, uint32_t idx, int lane) {
  const uint32_t L = uint32_t(lane);
  const uint32_t pos = a.local_pos[L & 15u];
  return ballot32((L < uint32_t(K)) & (((idx >> pos) & 1u) != 0u));
}

// The part of a tile's context that every workgroup of the pass shares (a kernel starts with it, so that the table
// lookup of thread_offsets travels with the per-bit tables), then the tile's own bits.
__device__ __forceinline__ TileCtx pass_tile_ctx(const PassArgs& a, const uint32_t* tables) {
  TileCtx t;
  t.tile_base = 0;
  t.c = a.c;
  t.cmask = (1u << a.c) - 1u;
  t.spread = tables + a.spread_off;
  t.shift = a.spread_shift;
#pragma unroll
  for (int i = 0; i < 8; ++i) t.ro[i] = a.row_off[i];
  return t;
}
__device__ __forceinline__ uint32_t tile_base_of(const PassArgs& a, uint32_t tile_id, int lane) {
  // tile-id bit i goes to the i-th nonlocal position (ascending): lane p deposits the bit of its rank
  const uint32_t L = uint32_t(lane) & 31u;
  const uint32_t rank = __popc(a.nonlocal_mask & ((1u << L) - 1u));
  return ballot32((lane < 32) & ((((a.nonlocal_mask >> L) & (tile_id >> rank)) & 1u) != 0u));
}

// Only the tiles that can hold a non-zero amplitude are launched (PassArgs::n_free): block b of the grid is
// tile `b & (2^n_free - 1)` of the LIVE tiles of state `b >> n_free`; the tile-id bits that belong to index
// bits of `zero_mask` are those of the input bitstring `idx` (given in the layout the pass loads).  Launching
// the dead tiles only to return cost 3 - 6 ms per pass at 4096 states (a million workgroups to dispatch).
__device__ __forceinline__ uint32_t launched_tile(const PassArgs& a, uint32_t block, uint32_t idx, int lane) {
  const uint32_t L = uint32_t(lane);
  const bool valid = L < a.n_nonlocal;
  const uint32_t pos = a.nonlocal_pos[L & 31u];
  const bool fixed = (a.zero_mask >> pos) & 1u;                     // this tile-id bit follows the input bitstring
  const uint32_t free_lanes = ballot32(valid & !fixed);             // the others take the bits of `live` in turn
  const uint32_t rank = __popc(free_lanes & ((1u << (L & 31u)) - 1u));
  const uint32_t live = block & ((1u << a.n_free) - 1u);
  const uint32_t bit = fixed ? (idx >> pos) & 1u : (live >> rank) & 1u;
  return ballot32(valid & (bit != 0u));
}

// Round geometry.  Thread `tid` owns the 2^R amplitudes whose local index has
// tid's bits deposited on the non-register positions (TL, read from the scheduler's per-round table);
// register value m adds the bits of m on the register positions.  In swizzled slot space both parts
// combine by XOR, so the 2^R slots are visited in Gray-code order with one
// v_xor per access: slot(gray(i)) = slot(gray(i-1)) ^ DB[ctz(i)].  Slots are kept
// as BYTE offsets (x8) so an access needs no further address arithmetic.
template <int K, int R>
__device__ __forceinline__ void round_geometry(uint32_t regmask, uint32_t tl, uint32_t (&DB)[R], uint32_t* T) {
  uint32_t mk = regmask;  // wave-uniform: scalar arithmetic
  DB[0] = swz(mk & (0u - mk)) << 3;  // lowest set bit, swizzled, in bytes
  mk &= mk - 1;
  DB[1] = swz(mk & (0u - mk)) << 3;
  mk &= mk - 1;
  DB[2] = swz(mk & (0u - mk)) << 3;
  mk &= mk - 1;
  DB[3] = swz(mk & (0u - mk)) << 3;
  if constexpr (R > 4) {
    mk &= mk - 1;
    DB[R > 4 ? 4 : 0] = swz(mk & (0u - mk)) << 3;
  }
  *T = swz(tl) << 3;
}

template <int R, int... I>
__device__ __forceinline__ void round_load_(const char* __restrict__ base, uint32_t addr,
                                            const uint32_t (&DB)[R], v2f (&a)[1 << R],
                                            std::integer_sequence<int, I...>) {
  ((addr ^= (I ? DB[I ? __builtin_ctz(I) : 0] : 0u),
    a[I ^ (I >> 1)] = *reinterpret_cast<const v2f*>(base + addr)), ...);
}
template <int R>
__device__ __forceinline__ void round_load(const float2* __restrict__ tile, uint32_t T,
                                           const uint32_t (&DB)[R], v2f (&a)[1 << R]) {
  round_load_<R>(reinterpret_cast<const char*>(tile), T, DB, a, iseq<(1 << R)>{});
}

template <int R, int... I>
__device__ __forceinline__ void round_store_(char* __restrict__ base, uint32_t addr,
                                             const uint32_t (&DB)[R], const v2f (&a)[1 << R],
                                             std::integer_sequence<int, I...>) {
  ((addr ^= (I ? DB[I ? __builtin_ctz(I) : 0] : 0u),
    *reinterpret_cast<v2f*>(base + addr) = a[I ^ (I >> 1)]), ...);
}
template <int R>
__device__ __forceinline__ void round_store(float2* __restrict__ tile, uint32_t T,
                                            const uint32_t (&DB)[R], const v2f (&a)[1 << R]) {
  round_store_<R>(reinterpret_cast<char*>(tile), T, DB, a, iseq<(1 << R)>{});
}

// The same exchange for a tile that starts at LDS byte 0 (the paired forward kernel and the exchange adjoint kernel: their
// only LDS is the dynamic array, checked once per workgroup by lds_tile_at_zero): the accesses go through ABSOLUTE LDS
// addresses.  `base + addr` on the array's symbol costs a v_add_u32 with a link-time 0 per access -- 2^R per exchange, none
// of which the compiler can fold (the XOR walk keeps `addr` out of the instruction's offset field).
typedef v2f __attribute__((address_space(3))) * LdsPair;
__device__ __forceinline__ void lds_tile_at_zero(const void* tile) {
  if (uint32_t(uintptr_t((__attribute__((address_space(3))) const char*)(tile))) != 0u) __builtin_trap();
}
template <int R, int... I>
__device__ __forceinline__ void round_load0_(uint32_t addr, const uint32_t (&DB)[R], v2f (&a)[1 << R],
                                             std::integer_sequence<int, I...>) {
  ((addr ^= (I ? DB[I ? __builtin_ctz(I) : 0] : 0u), a[I ^ (I >> 1)] = *reinterpret_cast<LdsPair>(uintptr_t(addr))), ...);
}
template <int R>
__device__ __forceinline__ void round_load0(uint32_t T, const uint32_t (&DB)[R], v2f (&a)[1 << R]) {
  round_load0_<R>(T, DB, a, iseq<(1 << R)>{});
}
template <int R, int... I>
__device__ __forceinline__ void round_store0_(uint32_t addr, const uint32_t (&DB)[R], const v2f (&a)[1 << R],
                                              std::integer_sequence<int, I...>) {
  ((addr ^= (I ? DB[I ? __builtin_ctz(I) : 0] : 0u), *reinterpret_cast<LdsPair>(uintptr_t(addr)) = a[I ^ (I >> 1)]), ...);
}
template <int R>
__device__ __forceinline__ void round_store0(uint32_t T, const uint32_t (&DB)[R], const v2f (&a)[1 << R]) {
  round_store0_<R>(T, DB, a, iseq<(1 << R)>{});
}

// Dense two-qubit gate applied directly on the LDS tile (not on the hot path of the
// hardware-efficient ansatz; keeps the register rounds free of 4x4 code).
// pos0/pos1 = local bit of the first/second qubit; matrix index = (b_q0 << 1) | b_q1.
__device__ __forceinline__ void quad_indices(uint32_t q, uint32_t pos0, uint32_t pos1, uint32_t (&ix)[4]) {
  const uint32_t pa = pos0 < pos1 ? pos0 : pos1, pb = pos0 < pos1 ? pos1 : pos0;
  uint32_t l = ((q >> pa) << (pa + 1)) | (q & ((1u << pa) - 1u));
  l = ((l >> pb) << (pb + 1)) | (l & ((1u << pb) - 1u));
#pragma unroll
  for (int j = 0; j < 4; ++j) ix[j] = swz(l | (uint32_t(j >> 1) << pos0) | (uint32_t(j & 1) << pos1));
}

__device__ __forceinline__ void mat4_apply(const float* __restrict__ u, const float2 (&x)[4], float2 (&y)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ur = u[(i * 4 + j) * 2], ui = u[(i * 4 + j) * 2 + 1];
      sr += ur * x[j].x - ui * x[j].y;
      si += ur * x[j].y + ui * x[j].x;
    }
    y[i] = make_float2(sr, si);
  }
}


// ---- fixed-layout instance records (program.h RecordLayout) -------------------------------
// rv[i] holds words 64*i .. 64*i+63 of the record, one word per lane.
template <int W, int NV>
__device__ __forceinline__ uint32_t rec_word(const uint32_t (&rv)[NV]) {
  return uint32_t(__builtin_amdgcn_readlane(int(rv[W / 64]), W % 64));
}
template <int W, int NV>
__device__ __forceinline__ v2f rec_cs(const uint32_t (&rv)[NV]) {
  return v2f{__uint_as_float(rec_word<W>(rv)), __uint_as_float(rec_word<W + 1>(rv))};
}
// The same fields straight from memory at a wave-uniform address `rb` = recs + rec_off: the compiler
// turns them into scalar loads through the constant cache (s_load_dword..x8, no VALU) -- with both
// sweeps VALU-issue bound this beats the v_readlane decode by 2 % (it did not in round 1, when the
// kernels still waited on memory); the coalesced vector fetch stays for the per-lane slot vector.
// -DQHBM_SCALAR_RECORDS=0 restores the v_readlane decode for A/B measurements.
#ifndef QHBM_OBS_NT_STORE
#define QHBM_OBS_NT_STORE 1
#endif
#ifndef QHBM_SCALAR_RECORDS
#define QHBM_SCALAR_RECORDS 1
#endif
struct RecBase { const uint32_t* p; };
template <int W, int NV>
__device__ __forceinline__ uint32_t rec_word(const uint32_t (&rv)[NV], RecBase rb) {
  if constexpr (QHBM_SCALAR_RECORDS) return rb.p[W];
  else return rec_word<W>(rv);
}
template <int W, int NV>
__device__ __forceinline__ v2f rec_cs(const uint32_t (&rv)[NV], RecBase rb) {
  return v2f{__uint_as_float(rec_word<W>(rv, rb)), __uint_as_float(rec_word<W + 1>(rv, rb))};
}
template <int NV>
__device__ __forceinline__ void rec_load(const uint32_t* __restrict__ recs, uint32_t off, int lane,
                                         uint32_t (&rv)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) rv[i] = recs[off + 64u * i + uint32_t(lane)];
}

// Gradient partial of one slot: wave sum (DPP rows + readlane).  Every wave executes every gate of
// the pass exactly once, so it OWNS one LDS cell per slot (cells[slot][wave]) and stores into it --
// no atomics; the kernel epilogue adds a slot's cells in wave order, so the tile's gradient is
// bit-identical from run to run.
template <int NW>
__device__ __forceinline__ void add_slot(float* cells, int tid, uint32_t slot, float v) {
  v = wave_sum(v);
  if ((tid & 63) == 0) cells[slot * NW + (uint32_t(tid) >> 6)] = v;
}

// Gradient partials of the EIGHT slots of record slot group G8 (program.h slot_lane8), reduced over the wave together
// WITHOUT selects.  Value v = (v0, v1, v2) ends up in the lanes whose bits (2, 3, 4) spell v (`present`: a wave-uniform
// bit per value that exists -- the instance's own micro-op masks; used by the A/B builds with presence tests only):
//   level 1  lane bit 2 (adjacent banks of four lanes): value 2k adds its partner bank's share under bank_mask 0x5
//            (row_shl:4), value 2k + 1 under 0xa (row_shr:4) into the SAME register -- one DPP add per value;
//   level 2  lane bit 3: pair (2k, 2k + 1) with pair (2k + 2, 2k + 3) under bank masks 0x3 / 0xc (row_shl:8 / row_shr:8)
//            -- one DPP add per pair;
//   level 3  v_permlane16_swap + add: values 0..3 stay in even rows, 4..7 in odd rows (summed over the row pair);
//   then the sums nobody selects on: two quad butterflies (lane bits 0, 1) and v_permlane32_swap + add (lane bit 5).
// Round 4's butterfly paid two v_cndmask_b32 (4.2 cycles each on gfx950, as much as a packed FMA:
// scripts/experiments/micro/valu_cycles.hip) per DPP add and all eight inputs whether they existed or not: 12 selects +
// 8 DPP + 2 swaps = 105 cycles per call; here 12 DPP adds + a 31-cycle tail = 81, and no zeroed inputs: an input that
// does not exist holds whatever its register held, and reaches no stored sum.  One asm statement: the DPP hazards (a VGPR
// written by the previous VALU instruction needs two wait states before a DPP read) are spelled out.
// The lanes whose slot-vector word `sv` IS the slot of the value they hold then store into their wave's cells (see
// add_slot).  (`sv` holds slots LOCAL to the pass; the chain-rule scale of the slot class is folded into slot_factor.)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int G8, int NW>
__device__ __forceinline__ void add_slots8(float* cells, int lane, uint32_t wave, uint32_t sv, uint32_t present, float g0,
                                           float g1, float g2, float g3, float g4, float g5, float g6, float g7) {
  float t0, t1, t2, t3, u0, u1, w, x;
// Presence tests (a scalar bit test + branch around the DPP add of a value or pair that does not exist) were measured
// and LOSE: a taken branch costs a wave more than the 4.2-cycle DPP add it skips (adjoint 230.9 / 229.7 / 228.5 ms with a
// test per value and pair / per pair / none, profiles/r05_ab_runs.txt).  0 = none (shipped); 1, 2: A/B builds.
#ifndef QHBM_RED_TESTS
#define QHBM_RED_TESTS 0
#endif
#if QHBM_RED_TESTS >= 2
#define QHBM_L1(BIT_, T_, G_, DIR_, BANKS_)                                                        \
  "s_bitcmp1_b32 %[m], " #BIT_ "\n\t"                                                              \
  "s_cbranch_scc0 .Lred%=_a" #BIT_ "\n\t"                                                          \
  "v_add_f32_dpp %[" #T_ "], %[" #G_ "], %[" #G_ "] " DIR_ ":4 row_mask:0xf bank_mask:" BANKS_ "\n" \
  ".Lred%=_a" #BIT_ ":\n\t"
#else
#define QHBM_L1(BIT_, T_, G_, DIR_, BANKS_)                                                        \
  "v_add_f32_dpp %[" #T_ "], %[" #G_ "], %[" #G_ "] " DIR_ ":4 row_mask:0xf bank_mask:" BANKS_ "\n\t"
#endif
#if QHBM_RED_TESTS >= 1
#define QHBM_L2(MASK_, TAG_, U_, T_, DIR_, BANKS_)                                                 \
  "s_and_b32 %[sc], %[m], " MASK_ "\n\t"                                                          \
  "s_cbranch_scc0 .Lred%=_b" #TAG_ "\n\t"                                                         \
  "v_add_f32_dpp %[" #U_ "], %[" #T_ "], %[" #T_ "] " DIR_ ":8 row_mask:0xf bank_mask:" BANKS_ "\n" \
  ".Lred%=_b" #TAG_ ":\n\t"
#else
#define QHBM_L2(MASK_, TAG_, U_, T_, DIR_, BANKS_)                                                 \
  "v_add_f32_dpp %[" #U_ "], %[" #T_ "], %[" #T_ "] " DIR_ ":8 row_mask:0xf bank_mask:" BANKS_ "\n\t"
#endif
  uint32_t sc;
  asm volatile(
      "s_nop 1\n\t"
      QHBM_L1(0, t0, g0, "row_shl", "0x5") QHBM_L1(2, t1, g2, "row_shl", "0x5") QHBM_L1(4, t2, g4, "row_shl", "0x5")
      QHBM_L1(6, t3, g6, "row_shl", "0x5") QHBM_L1(1, t0, g1, "row_shr", "0xa") QHBM_L1(3, t1, g3, "row_shr", "0xa")
      QHBM_L1(5, t2, g5, "row_shr", "0xa") QHBM_L1(7, t3, g7, "row_shr", "0xa")
      "s_nop 1\n\t"
      QHBM_L2("0x03", 0, u0, t0, "row_shl", "0x3") QHBM_L2("0x30", 2, u1, t2, "row_shl", "0x3")
      QHBM_L2("0x0c", 1, u0, t1, "row_shr", "0xc") QHBM_L2("0xc0", 3, u1, t3, "row_shr", "0xc")
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
      : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [w] "=&v"(w),
        [x] "=&v"(x), [sc] "=&s"(sc)
      : [m] "s"(present), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [g3] "v"(g3), [g4] "v"(g4), [g5] "v"(g5), [g6] "v"(g6),
        [g7] "v"(g7)
      : "scc");
#undef QHBM_L1
#undef QHBM_L2
  if ((lane & 3) == (G8 & 3) && (lane >> 5) == (G8 >> 2) && sv != 0xffffffffu) cells[sv * NW + wave] = w;
}

// A float nobody has written: an input of the eight-wide reduction whose slot does not exist.  The butterfly never adds
// DIFFERENT values (each level selects, then adds the same value of the partner lane), so what an unused input holds
// reaches no stored sum -- and a zero would cost a v_mov_b32 per input and instance (4 % of the adjoint's instructions).
__device__ __forceinline__ float any_float() {
  float v;
  asm volatile("" : "=v"(v));  // (volatile: one register per use -- a shared one would be copied into every input)
  return v;
}

// FULL diagonal table: amplitude with register value m (1..15) times FULL[m-1].
template <int NV, int... M>
__device__ __forceinline__ void apply_full_(v2f (&a)[16], const uint32_t (&rv)[NV], RecBase rb, bool conj,
                                            std::integer_sequence<int, M...>) {
  constexpr RecordLayout L(4, false);
  (phase_s(a[M + 1], conj ? conj_cs(rec_cs<L.full(M + 1)>(rv, rb)) : rec_cs<L.full(M + 1)>(rv, rb)), ...);
}
template <int NV>
__device__ __forceinline__ void apply_full(v2f (&a)[16], const uint32_t (&rv)[NV], RecBase rb, bool conj) {
  apply_full_<NV>(a, rv, rb, conj, iseq<15>{});
}
// q[m] = (lam.re * psi.im, lam.im * psi.re): Im(conj(lam) psi) = q.x - q.y, subtracted once per SUM, not
// per amplitude -- eight packed multiplies per asm statement (plain C++ here makes the SLP vectoriser
// gather the scalars with 32 v_mov before it packs them).
__device__ __forceinline__ void q8(v2f& q0, v2f& q1, v2f& q2, v2f& q3, v2f& q4, v2f& q5, v2f& q6, v2f& q7, v2f l0, v2f p0,
                                   v2f l1, v2f p1, v2f l2, v2f p2, v2f l3, v2f p3, v2f l4, v2f p4, v2f l5, v2f p5, v2f l6,
                                   v2f p6, v2f l7, v2f p7) {
#define QHBM_Q(K_) "v_pk_mul_f32 %[q" #K_ "], %[l" #K_ "], %[p" #K_ "] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
  asm(QHBM_Q(0) QHBM_Q(1) QHBM_Q(2) QHBM_Q(3) QHBM_Q(4) QHBM_Q(5) QHBM_Q(6)
      "v_pk_mul_f32 %[q7], %[l7], %[p7] op_sel:[0,1] op_sel_hi:[1,0]"
      : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3), [q4] "=&v"(q4), [q5] "=&v"(q5), [q6] "=&v"(q6),
        [q7] "=&v"(q7)
      : [l0] "v"(l0), [p0] "v"(p0), [l1] "v"(l1), [p1] "v"(p1), [l2] "v"(l2), [p2] "v"(p2), [l3] "v"(l3), [p3] "v"(p3),
        [l4] "v"(l4), [p4] "v"(p4), [l5] "v"(l5), [p5] "v"(p5), [l6] "v"(l6), [p6] "v"(p6), [l7] "v"(l7), [p7] "v"(p7));
#undef QHBM_Q
}
// d_k = Im(conj(lam_k) psi_k) = lam.re psi.im - lam.im psi.re for four amplitudes, as SCALARS: the sums below need the
// difference only, and a plain 32-bit VALU op issues in 2.2 cycles against 4.2 for a packed one (valu_cycles.hip) -- the
// packed form (q = (lam.re psi.im, lam.im psi.re), subtract last) carried both halves through 28 packed adds.
__device__ __forceinline__ void im4(float& d0, float& d1, float& d2, float& d3, v2f l0, v2f p0, v2f l1, v2f p1, v2f l2, v2f p2,
                                    v2f l3, v2f p3) {
  asm("v_mul_f32 %[d0], %[l0x], %[p0y]\n\t"
      "v_mul_f32 %[d1], %[l1x], %[p1y]\n\t"
      "v_mul_f32 %[d2], %[l2x], %[p2y]\n\t"
      "v_mul_f32 %[d3], %[l3x], %[p3y]\n\t"
      "v_fma_f32 %[d0], -%[l0y], %[p0x], %[d0]\n\t"
      "v_fma_f32 %[d1], -%[l1y], %[p1x], %[d1]\n\t"
      "v_fma_f32 %[d2], -%[l2y], %[p2x], %[d2]\n\t"
      "v_fma_f32 %[d3], -%[l3y], %[p3x], %[d3]"
      : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
      : [l0x] "v"(l0.x), [l0y] "v"(l0.y), [p0x] "v"(p0.x), [p0y] "v"(p0.y), [l1x] "v"(l1.x), [l1y] "v"(l1.y), [p1x] "v"(p1.x),
        [p1y] "v"(p1.y), [l2x] "v"(l2.x), [l2y] "v"(l2.y), [p2x] "v"(p2.x), [p2y] "v"(p2.y), [l3x] "v"(l3.x), [l3y] "v"(l3.y),
        [p3x] "v"(p3.x), [p3y] "v"(p3.y));
}
// Gradient partials of ALL one- and two-bit phase terms on the four register bits at once (FULL instances): with
// d[m] = Im(conj(lam_m) psi_m), A[k] = d[2k] + d[2k+1] and B[k] = d[2k+1] (k = register bits 1..3), the superset sums
// (z[k] <- sum of z[k'] over k' containing k) of B are the terms that contain bit 0 and those of A the ones that do not:
// 28 scalar adds for the ten sums, in place, in ONE asm statement (plain C++ here is SLP-packed into v_pk_add_f32 with a
// v_mov per operand).  g1[J] = PH1 on bit J, g2[pair_index(JA, JB)] = PH2 on (JA, JB); sums of terms the instance does
// not have are computed too and never stored (their slot is 0xffffffff).
__device__ __forceinline__ void full_partials(const v2f (&p)[16], const v2f (&l)[16], float (&g1)[4], float (&g2)[6]) {
  float d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11, d12, d13, d14, d15;
  im4(d0, d1, d2, d3, l[0], p[0], l[1], p[1], l[2], p[2], l[3], p[3]);
  im4(d4, d5, d6, d7, l[4], p[4], l[5], p[5], l[6], p[6], l[7], p[7]);
  im4(d8, d9, d10, d11, l[8], p[8], l[9], p[9], l[10], p[10], l[11], p[11]);
  im4(d12, d13, d14, d15, l[12], p[12], l[13], p[13], l[14], p[14], l[15], p[15]);
  // A[k] lives in d[2k], B[k] in d[2k+1].  Not needed and not computed: A[0] (the sum over everything), A[7] / B[3] /
  // B[5] / B[6] / B[7] as RESULTS (three-bit terms do not exist) -- they still feed the others.
#define QHBM_ADD(X_, Y_) "v_add_f32 %[d" #X_ "], %[d" #X_ "], %[d" #Y_ "]\n\t"
  asm(// A[k] = d[2k] + d[2k+1]
      QHBM_ADD(2, 3) QHBM_ADD(4, 5) QHBM_ADD(6, 7) QHBM_ADD(8, 9) QHBM_ADD(10, 11) QHBM_ADD(12, 13) QHBM_ADD(14, 15)
      // zeta over A (indices 2k), level k bit 0, 1, 2 -- without the chain into A[0]
      QHBM_ADD(4, 6) QHBM_ADD(8, 10) QHBM_ADD(12, 14)      // A2 += A3, A4 += A5, A6 += A7
      QHBM_ADD(2, 6) QHBM_ADD(8, 12) QHBM_ADD(10, 14)      // A1 += A3, A4 += A6, A5 += A7
      QHBM_ADD(2, 10) QHBM_ADD(4, 12) QHBM_ADD(6, 14)      // A1 += A5, A2 += A6, A3 += A7
      // zeta over B (indices 2k + 1)
      QHBM_ADD(1, 3) QHBM_ADD(5, 7) QHBM_ADD(9, 11) QHBM_ADD(13, 15)   // B0 += B1, B2 += B3, B4 += B5, B6 += B7
      QHBM_ADD(1, 5) QHBM_ADD(3, 7) QHBM_ADD(9, 13) QHBM_ADD(11, 15)   // B0 += B2, B1 += B3, B4 += B6, B5 += B7
      QHBM_ADD(1, 9) QHBM_ADD(3, 11) "v_add_f32 %[d5], %[d5], %[d13]"  // B0 += B4, B1 += B5, B2 += B6
      : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [d4] "+v"(d4), [d5] "+v"(d5), [d6] "+v"(d6), [d7] "+v"(d7),
        [d8] "+v"(d8), [d9] "+v"(d9), [d10] "+v"(d10), [d11] "+v"(d11), [d12] "+v"(d12), [d13] "+v"(d13), [d14] "+v"(d14),
        [d15] "+v"(d15));
#undef QHBM_ADD
  g1[0] = d1;   // B0: every amplitude with bit 0
  g1[1] = d2;   // A1
  g1[2] = d4;   // A2
  g1[3] = d8;   // A4
  g2[0] = d3;   // B1: bits (0, 1)
  g2[1] = d5;   // B2: (0, 2)
  g2[2] = d6;   // A3: (1, 2)
  g2[3] = d9;   // B4: (0, 3)
  g2[4] = d10;  // A5: (1, 3)
  g2[5] = d12;  // A6: (2, 3)
}

// Controlled phase: register bit J AND (a thread bit | a tile bit).  One code path for both
// predicate kinds: `tlx` = the thread's local index | the tile id << K (program.h CPHPRED), so a predicate is one
// bit of one word (a select between the two sources compiled to a VCC v_cndmask_b32: 23 cycles per wave on gfx950
// against 2.2 for a plain 32-bit VALU op -- scripts/experiments/micro/valu_cycles.hip); the
// phase degenerates to 1 where the predicate is false.  The scheduler maps the predicate bits of a
// round to WAVE bits where it can, so most predicates are uniform over a wave and half the waves
// skip the micro-op altogether.
template <int R, int J>
__device__ __forceinline__ void cph_fwd(v2f (&a)[1 << R], v2f cs, uint32_t pred, uint32_t tlx) {
  const bool on = (tlx >> (pred & 0x1fu)) & 1u;
  if (__builtin_amdgcn_ballot_w64(on) == 0) return;  // off in the whole wave (tile bit, or a wave bit: schedule.cpp emit_round)
  if (on) apply_ph1_s8<R, J>(a, cs);
}

template <int R, int J>
__device__ __forceinline__ float cph_adj(v2f (&p)[1 << R], v2f (&l)[1 << R], v2f cs, uint32_t pred,
                                         uint32_t tlx) {
  const bool on = (tlx >> (pred & 0x1fu)) & 1u;
  if (__builtin_amdgcn_ballot_w64(on) == 0) return 0.f;  // off in the whole wave: nothing to do
  float g = 0.f;
  if (on) {
    g = sum_w1<R, J>(p, l);
    apply_ph1_s8<R, J>(p, cs);  // (cs: the conjugate phase, as the adjoint records hold it)
    apply_ph1_s8<R, J>(l, cs);
  }
  return g;
}

// One forward instance on the register file.
template <int R, int NV, bool GEN>
__device__ __forceinline__ void instance_fwd(const uint32_t (&rv)[NV], const uint32_t* __restrict__ recs,
                                             uint32_t rec_off, int lane, v2f (&a)[1 << R], uint32_t tlx) {
  constexpr RecordLayout L(R, false);
  const RecBase rb{recs + rec_off};
  const uint32_t h0 = rec_word<0>(rv, rb), h1 = rec_word<1>(rv, rb);
  // One-qubit gates: a separate predicated slot class per kind (X, Y, dense), each a plain
  // if-then triangle around in-place code -- no merge copies.
  QHBM_FOR_RB(R, if ((h0 >> J) & 1u) apply_x<R, J>(a, rec_cs<L.x(J)>(rv, rb));)
  if constexpr (GEN) {
  if ((h1 >> 16) & 0xfu) {
    QHBM_FOR_RB(R, if ((h1 >> (16 + J)) & 1u) apply_y<R, J>(a, rec_cs<L.y(J)>(rv, rb));)
  }
  if ((h1 >> 24) & 0xfu) {  // dense 2x2 gates (rare): their coefficients sit in the record's third part
    uint32_t dv[1];
    rec_load<1>(recs, rec_off + 128u, lane, dv);
    QHBM_FOR_RB(R,
      if ((h1 >> (24 + J)) & 1u)
        apply_mat1<R, J>(a, rec_cs<8 * J>(dv), rec_cs<8 * J + 2>(dv), rec_cs<8 * J + 4>(dv), rec_cs<8 * J + 6>(dv));)
  }
  }
  // (a FULL instance has its PH1 / PH2 masks zeroed in word 0: independent triangles, no else)
  if (h1 & kFullDiagFlag) apply_full<NV>(a, rv, rb, false);
  QHBM_FOR_RB(R, if ((h0 >> (8 + J)) & 1u) apply_ph1<R, J>(a, rec_cs<L.ph1(J)>(rv, rb));)
  if ((h0 >> 16) & 0x3fu) {
    QHBM_FOR_PAIR(R,
      if ((h0 >> (16 + pair_index(JA, JB))) & 1u) apply_ph2<R, JA, JB>(a, rec_cs<L.ph2(pair_index(JA, JB))>(rv, rb));)
  }
  if (h1 & 0xffu) {
    QHBM_FOR_RB(R,
      if ((h1 >> (2 * J)) & 1u)
        cph_fwd<R, J>(a, rec_cs<L.cph(2 * J)>(rv, rb), rec_word<L.pred(2 * J)>(rv, rb), tlx);
      if ((h1 >> (2 * J + 1)) & 1u)
        cph_fwd<R, J>(a, rec_cs<L.cph(2 * J + 1)>(rv, rb), rec_word<L.pred(2 * J + 1)>(rv, rb), tlx);)
  }
}

// The same instance on TWO register files (pass_fwd2_kernel: the same tile of two states): every micro-op's
// scalar predicate, record fields and branch are paid once for both.  Lean programs only.
template <int R, int J>
__device__ __forceinline__ void cph_fwd2(v2f (&a)[1 << R], v2f (&b)[1 << R], v2f cs, uint32_t pred, uint32_t tlx) {
  const bool on = (tlx >> (pred & 0x1fu)) & 1u;
  if (__builtin_amdgcn_ballot_w64(on) == 0) return;
  if (on) {
    apply_ph1_s8<R, J>(a, cs);
    apply_ph1_s8<R, J>(b, cs);
  }
}
template <int R, int NV>
__device__ __forceinline__ void instance_fwd_pair(const uint32_t (&rv)[NV], const uint32_t* __restrict__ recs,
                                                  uint32_t rec_off, v2f (&a)[1 << R], v2f (&b)[1 << R], uint32_t tlx) {
  constexpr RecordLayout L(R, false);
  const RecBase rb{recs + rec_off};
  const uint32_t h0 = rec_word<0>(rv, rb), h1 = rec_word<1>(rv, rb);
  QHBM_FOR_RB(R, if ((h0 >> J) & 1u) { const v2f cs = rec_cs<L.x(J)>(rv, rb); apply_x<R, J>(a, cs); apply_x<R, J>(b, cs); })
  if (h1 & kFullDiagFlag) { apply_full<NV>(a, rv, rb, false); apply_full<NV>(b, rv, rb, false); }
  QHBM_FOR_RB(R, if ((h0 >> (8 + J)) & 1u) { const v2f cs = rec_cs<L.ph1(J)>(rv, rb); apply_ph1<R, J>(a, cs); apply_ph1<R, J>(b, cs); })
  if ((h0 >> 16) & 0x3fu) {
    QHBM_FOR_PAIR(R,
      if ((h0 >> (16 + pair_index(JA, JB))) & 1u) {
        const v2f cs = rec_cs<L.ph2(pair_index(JA, JB))>(rv, rb);
        apply_ph2<R, JA, JB>(a, cs);
        apply_ph2<R, JA, JB>(b, cs);
      })
  }
  if (h1 & 0xffu) {
    QHBM_FOR_RB(R,
      if ((h1 >> (2 * J)) & 1u)
        cph_fwd2<R, J>(a, b, rec_cs<L.cph(2 * J)>(rv, rb), rec_word<L.pred(2 * J)>(rv, rb), tlx);
      if ((h1 >> (2 * J + 1)) & 1u)
        cph_fwd2<R, J>(a, b, rec_cs<L.cph(2 * J + 1)>(rv, rb), rec_word<L.pred(2 * J + 1)>(rv, rb), tlx);)
  }
}

// Measurement helpers (register file indexed by the high bits of the local index).
__device__ __forceinline__ v2f meas_w(const float2* __restrict__ tile, uint32_t s, uint32_t xs) {
  const float2 p = tile[s];
  const float2 q = tile[s ^ xs];  // psi[l ^ x]
  return v2f{q.x * p.x + q.y * p.y, q.x * p.y - q.y * p.x};
}
template <int R, int NT, int... I>
__device__ __forceinline__ void meas_load_(const float2* __restrict__ tile, uint32_t tid, uint32_t xs,
                                           v2f (&w)[1 << R], std::integer_sequence<int, I...>) {
  ((w[I] = meas_w(tile, swz(uint32_t(I) * NT + tid), xs)), ...);
}
template <int R, int NT>
__device__ __forceinline__ void meas_load(const float2* __restrict__ tile, uint32_t tid, uint32_t xs,
                                          v2f (&w)[1 << R]) {
  meas_load_<R, NT>(tile, tid, xs, w, iseq<(1 << R)>{});
}
template <int R, int IM, int... I>
__device__ __forceinline__ float meas_sum_(const v2f (&w)[1 << R], uint32_t zhi, std::integer_sequence<int, I...>) {
  return (((__builtin_popcount(uint32_t(I) & zhi) & 1) ? -(IM ? w[I].y : w[I].x) : (IM ? w[I].y : w[I].x)) + ...);
}
template <int R, int IM>
__device__ __forceinline__ float meas_sum(const v2f (&w)[1 << R], uint32_t zhi) {
  return meas_sum_<R, IM>(w, zhi, iseq<(1 << R)>{});
}

// ---- relabeling adjoint plans (schedule.h Pass, program.h PassArgs) ------------------------------------
// Index bits finished by EARLIER passes that this tile holds as local bits: where they differ from the
// input the memory holds stale data (the finishing pass stored the live half only) -- psi is zero there
// and lambda there is never needed again, so the prefetched amplitudes are cleared.
template <int K>
__device__ __forceinline__ void clear_stale(TileRegs& r, int tid, uint32_t in_local, uint32_t fz) {
#define QHBM_CL(I)                                                                   \
  {                                                                                  \
    const uint32_t l = 2u * uint32_t(tid) + (uint32_t(I) << (K - 3));                \
    if ((l ^ in_local) & fz) { r.p##I.x = 0.f; r.p##I.y = 0.f; }                     \
    if (((l | 1u) ^ in_local) & fz) { r.p##I.z = 0.f; r.p##I.w = 0.f; }              \
  }
  QHBM_CL(0) QHBM_CL(1) QHBM_CL(2) QHBM_CL(3) QHBM_CL(4) QHBM_CL(5) QHBM_CL(6) QHBM_CL(7)
#undef QHBM_CL
}
// The value of lane (l ^ 2^B) of the wave, for the butterflies of measure_diagonal_wht: quad permutes, LDS-crossbar
// swizzles (no LDS memory involved) and, across the two halves, a bpermute.
template <int B> __device__ __forceinline__ float lane_xor(float v, int lane) {
  if constexpr (B == 0) return dpp_get<0xB1>(v);        // quad_perm:[1,0,3,2]
  else if constexpr (B == 1) return dpp_get<0x4E>(v);   // quad_perm:[2,3,0,1]
  else if constexpr (B <= 4) return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), ((1 << B) << 10) | 0x1f));
  else return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v)));
}
template <int B, int NR> __device__ __forceinline__ void wht_lane_stage(float (&w)[NR], int lane) {
  const float s = (lane >> B) & 1 ? -1.f : 1.f;  // lane bit clear: w + partner, set: partner - w
#pragma unroll
  for (int i = 0; i < NR; ++i) w[i] = fmaf(s, w[i], lane_xor<B>(w[i], lane));
}

// OP_MEASURE_WHT (program.h): every diagonal term of the group from ONE Walsh-Hadamard transform of the tile's
// probabilities.  Thread tid holds p[i] = |psi[l]|^2 at l = i NT + tid; after the transform over the register index
// i (in registers) and the six lane bits (cross-lane), w[c] in lane L is  sum_{i, j} (-1)^{popc(i & c) + popc(j & L)}
// p over the wave's 64 NR amplitudes: the term with local z mask zl is w[zl >> (K - R)] of lane zl & 63, times the
// parity of its wave bits and of its tile bits -- one LDS gather per term, the terms dealt to the lanes, every lane
// adding its term to the (integer) accumulator of the term's operator.  `scr`: 64 floats private to the wave.
template <int K, int R, int NT>
__device__ __forceinline__ void measure_diagonal_wht(const float2* __restrict__ tile, int tid, uint32_t tile_base,
                                                     const uint32_t* __restrict__ class_end,
                                                     const uint32_t* __restrict__ terms, float* scr,
                                                     const float* __restrict__ op_scale, unsigned long long* red) {
  constexpr int NR = 1 << R;
  const int lane = tid & 63;
  const uint32_t wave = uint32_t(tid) >> 6;
  float w[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const float2 v = tile[swz(uint32_t(i) * NT + uint32_t(tid))];
    w[i] = v.x * v.x + v.y * v.y;
  }
#pragma unroll
  for (int s = 0; s < R; ++s)
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (!(i >> s & 1)) {
        const float x = w[i], y = w[i | (1 << s)];
        w[i] = x + y;
        w[i | (1 << s)] = x - y;
      }
  wht_lane_stage<0>(w, lane);
  wht_lane_stage<1>(w, lane);
  wht_lane_stage<2>(w, lane);
  wht_lane_stage<3>(w, lane);
  wht_lane_stage<4>(w, lane);
  wht_lane_stage<5>(w, lane);
  uint32_t begin = 0;
#pragma unroll
  for (int c = 0; c < NR; ++c) {
    const uint32_t end = uni(class_end[c]);
    if (end != begin) {  // (wave-uniform)
      scr[lane] = w[c];  // the wave's own 64 floats: DS operations of a wave execute in order
      for (uint32_t k = begin + uint32_t(lane); k < end; k += 64u) {
        const uint32_t* tw = terms + size_t(k) * kMeasTermWords;  // (the program is only 4-byte aligned)
        const uint4 t = make_uint4(tw[0], tw[1], tw[2], tw[3]);    // zl, zn, coefficient, operator
        const uint32_t par = uint32_t(__popc(tile_base & t.y) + __popc(wave & (t.x >> 6))) & 1u;
        const float v = __uint_as_float(t.z ^ (par << 31)) * scr[t.x & 63u];
        atomicAdd(&red[t.w], to_fixed(v, op_scale[t.w]));
      }
    }
    begin = end;
  }
}

}  // namespace

// ================================================================================
// Forward pass kernel
// ================================================================================
template <int K, int R, bool GEN>
__global__ __launch_bounds__(1 << (K - R), fwd_min_waves(K, R)) void pass_fwd_kernel(
    PassArgs a, float2* __restrict__ psi, const int8_t* __restrict__ bits, int n_user,
    const uint32_t* __restrict__ prog_base, const uint32_t* __restrict__ tables,
    const float* __restrict__ coef, const float* __restrict__ op_scale, unsigned long long* __restrict__ out64,
    uint32_t state0, const float2* __restrict__ psi_src) {
  constexpr int NT = 1 << (K - R);
  constexpr int NR = 1 << R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tile = reinterpret_cast<float2*>(smem);
  // per-op accumulators of this tile, 64-bit fixed point (program.h kValueFracBits): integer adds
  // commute, so neither the order of the waves here nor that of the tiles below changes a bit
  unsigned long long* red = reinterpret_cast<unsigned long long*>(tile + (1 << K));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const uint32_t s_local = blockIdx.x >> a.n_free;
  uint32_t bits_row = state0 + s_local, out_row = state0 + s_local;
  if (a.prog_states) {  // batched programs (PassArgs::prog_states): own coefficients, shared bitstrings
    const uint32_t q = s_local / a.prog_states;
    bits_row = state0 + (s_local - q * a.prog_states);
    out_row = s_local;
    coef += size_t(q) * a.coef_stride;
  }
  const uint32_t* recs = reinterpret_cast<const uint32_t*>(coef);
  TileCtx t = pass_tile_ctx(a, tables);
  const ThreadOff toff = thread_offsets(t, tid);
  uint32_t idx = 0;  // the input bitstring as an index: only the passes that prune or initialise read it
  if ((a.flags & PASS_INIT_BASIS) | a.zero_mask | a.frozen_old_local) idx = input_index(a, bits + size_t(bits_row) * n_user, n_user, lane);
  const uint32_t in_local = local_bits<K>(a, idx, lane);
  const uint32_t tile_id = launched_tile(a, blockIdx.x, idx, lane);
  t.tile_base = tile_base_of(a, tile_id, lane);
  const uint32_t tile_hi = tile_id << K;  // tile-bit predicates of boundary phases (cph_*): bit K + i = tile-id bit i
  float2* st = psi + (size_t(s_local) << a.n);

  if (a.flags & PASS_INIT_BASIS) {
    if ((idx & a.nonlocal_mask) != t.tile_base) {
      // The basis amplitude lives in another tile: this one is zero and stays zero under the
      // program (every op is linear), so only its image in HBM has to be written.
      // (PASS_NO_ZERO_FILL: nothing is written -- later passes never read these tiles unmasked, see schedule.cpp)
      if ((a.flags & PASS_STORE) && !(a.flags & PASS_NO_ZERO_FILL)) {
        for (int p = tid; p < (1 << (K - 1)); p += NT)
          *reinterpret_cast<float4*>(st + global_index(t, 2u * p)) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
    for (int l = tid; l < (1 << K); l += NT) tile[l] = make_float2(0.f, 0.f);
    __syncthreads();
    if (tid == 0) tile[swz(in_local)] = make_float2(1.f, 0.f);
  } else {
    // (head of the sweep: the tiles on which psi is identically zero are not launched -- launched_tile)
    TileRegs r;
    // (psi_src: the first pass of a shifted program that shares its prefix with the base program loads the BASE state of
    // its bitstring -- the passes before it are the same bits for every program -- and stores into its own element)
    const float2* ld = psi_src ? psi_src + (size_t(bits_row - state0) << a.n) : st;
    prefetch_tile<K, NT>(r, ld, t, toff);
    if (a.frozen_old_local) clear_stale<K>(r, tid, in_local, a.frozen_old_local);  // local bits nothing has acted on yet: their != input half was never written
    commit_tile<K, NT>(tile, r, tid);
  }
  for (int i = tid; i < kMaxOps; i += NT) red[i] = 0ull;
  __syncthreads();

  const uint32_t* prog = prog_base + a.prog_off;
  uint32_t pc = 0;
  constexpr int NV = 1;
  uint32_t cur[NV], nxt[NV];
  uint32_t carried_off = 0xffffffffu;  // record offset whose words `cur` holds
  for (;;) {
    const uint32_t w0 = uni(prog[pc]);
    const uint32_t opc = w0 & 0xffu;
    if (opc == OP_END) break;
    if (opc == OP_ROUND) {
      const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8;
      const uint32_t regmask = uni(prog[pc + 1]);
      uint32_t rec_off = uni(prog[pc + 2]);
      constexpr RecordLayout L(R, false);
      // The records of consecutive rounds are consecutive in the coefficient buffer, so the
      // one-ahead prefetch of the previous round's last instance already holds this round's first
      // record: reloading it would put a global-load latency in front of every round.
      if (rec_off != carried_off) rec_load<NV>(recs, rec_off, lane, cur);
      uint32_t DB[R], T;
      const uint32_t TL = tables[a.tl_off + uni(prog[pc + 3]) + uint32_t(tid)];
      round_geometry<K, R>(regmask, TL, DB, &T);
      v2f amp[NR];
      round_load<R>(tile, T, DB, amp);
      for (uint32_t i = 0; i < n_inst; ++i) {
        rec_load<NV>(recs, rec_off + L.words(), lane, nxt);  // prefetch (the buffer is padded)
        instance_fwd<R, NV, GEN>(cur, recs, rec_off, lane, amp, TL | tile_hi);
        rec_off += L.words();
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[v] = nxt[v];
      }
      carried_off = rec_off;
      round_store<R>(tile, T, DB, amp);
      if (!(w0 & kRoundNoBarrier)) __syncthreads();  // else the next round's waves read only their own writes
      pc += kRoundWords;
    } else if (opc == OP_GATE2) {
      if constexpr (GEN) {
        const uint32_t pw = uni(prog[pc + 1]);
        const uint32_t pos0 = pw & 0xffu, pos1 = (pw >> 8) & 0xffu;
        const float* cf = coef + uni(prog[pc + 2]);
        for (uint32_t q = tid; q < (1u << (K - 2)); q += NT) {
          uint32_t ix[4];
          quad_indices(q, pos0, pos1, ix);
          float2 x[4], y[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) x[j] = tile[ix[j]];
          mat4_apply(cf, x, y);
#pragma unroll
          for (int j = 0; j < 4; ++j) tile[ix[j]] = y[j];
        }
        __syncthreads();
      }
      pc += kGate2Words;
    } else if (opc == OP_MEASURE_WHT) {
      const uint32_t n_terms = w0 >> 8;
      const uint32_t* class_end = prog + pc + 1;
      const uint32_t* terms = prog + pc + kWhtHeaderWords;
      pc += uint32_t(kWhtHeaderWords) + n_terms * uint32_t(kMeasTermWords);
      if (!(a.flags & PASS_SKIP_MEASURE)) {
        // (scratch: the top NT floats of the accumulator array -- the scheduler emits this op only if they are free)
        float* scr = reinterpret_cast<float*>(red + kMaxOps) - NT + (tid & ~63);
        measure_diagonal_wht<K, R, NT>(tile, tid, t.tile_base, class_end, terms, scr, op_scale, red);
        __syncthreads();
      }
    } else {  // OP_MEASURE
      const uint32_t n_groups = w0 >> 8;
      pc += 1;
      if (a.flags & PASS_SKIP_MEASURE) {  // the values come from lambda = O psi (engine.cpp): walk over the groups
        for (uint32_t g = 0; g < n_groups; ++g) pc += 2u + uni(prog[pc + 1]) * kMeasTermWords;
        continue;
      }
      float acc = 0.f;
      uint32_t cur_op = 0xffffffffu;
      for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t xl = uni(prog[pc]), n_terms = uni(prog[pc + 1]);
        pc += 2;
        v2f w[NR];  // w = conj(psi[l ^ x]) * psi[l] at l = i*NT + tid
        meas_load<R, NT>(tile, uint32_t(tid), swz(xl), w);
        uint32_t last_key = 0xffffffffu;  // (zhi, real/imag) of the signed register sum held in `sum`
        float sum = 0.f;
        for (uint32_t k = 0; k < n_terms; ++k, pc += kMeasTermWords) {
          const uint32_t zl = uni(prog[pc]), zn = uni(prog[pc + 1]);
          const float cf = __uint_as_float(uni(prog[pc + 2]));
          const uint32_t ow = uni(prog[pc + 3]);
          const uint32_t op = ow & 0xffffffu, ny = ow >> 24;
          if (op != cur_op) {
            if (cur_op != 0xffffffffu) {
              const float v = wave_sum(acc);
              if ((tid & 63) == 0) atomicAdd(&red[cur_op], to_fixed(v, op_scale[cur_op]));
            }
            acc = 0.f;
            cur_op = op;
          }
          // Re( i^ny * (-1)^{popc(l & z)} * w ):  ny=0: wr, 1: -wi, 2: -wr, 3: wi
          float sfac = (ny == 1 || ny == 2) ? -cf : cf;
          if (__popc(t.tile_base & zn) & 1) sfac = -sfac;
          if (__popc(uint32_t(tid) & zl) & 1) sfac = -sfac;
          const uint32_t zhi = zl >> (K - R);  // bits of l above the thread index
          const uint32_t key = zhi | ((ny & 1u) << 16);
          if (key != last_key) {  // terms of a group mostly differ in thread / tile bits only
            sum = (ny & 1) ? meas_sum<R, 1>(w, zhi) : meas_sum<R, 0>(w, zhi);
            last_key = key;
          }
          acc = fmaf(sfac, sum, acc);
        }
      }
      if (cur_op != 0xffffffffu) {
        const float v = wave_sum(acc);
        if ((tid & 63) == 0) atomicAdd(&red[cur_op], to_fixed(v, op_scale[cur_op]));
      }
      __syncthreads();
    }
  }

  if (a.n_ops) {
    __syncthreads();
    for (uint32_t i = tid; i < a.n_ops; i += NT) {
      const unsigned long long v = red[i];
      if (v) atomicAdd(&out64[size_t(out_row) * a.n_ops + i], v);
    }
  }
  if (a.flags & PASS_STORE) store_tile<K, NT>(tile, st, t, thread_offsets(t, tid), tid);
}

// ================================================================================
// Forward pass on a PAIR of states (dense, lean, measurement-free passes): the exchange layout of the
// adjoint kernel below applied to the forward sweep.  A workgroup holds the same tile of two consecutive
// states in REGISTERS (16 + 16 amplitudes per thread) and runs every instance on both with one set of
// scalar record fields; LDS is one tile-sized exchange buffer through which first one state, then the
// other changes geometry between rounds.  Twice the packed-fp32 work per round trip, barrier, record
// decode and thread -> index table as pass_fwd_kernel: that kernel sat at 0.70 of the VALU issue rate
// (its waves wait on the LDS round trip between rounds), the adjoint kernel with this structure at 0.92.
// Used when nothing distinguishes the two states' tiles: no tile pruned, nothing stale, no measurement
// in the pass (or the values come from lambda = O psi), one program (engine.cpp run_forward_chunk).
// ================================================================================
template <int K>
__global__ __launch_bounds__(1 << (K - 4), adjx_min_waves(K)) void pass_fwd2_kernel(
    PassArgs a, float2* __restrict__ psi, const int8_t* __restrict__ bits, int n_user, uint32_t state0,
    const uint32_t* __restrict__ prog_base, const uint32_t* __restrict__ tables, const float* __restrict__ coef,
    uint32_t n_states) {
  constexpr int R = 4;
  constexpr int NT = 1 << (K - R);
  constexpr int NR = 1 << R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* xt = reinterpret_cast<float2*>(smem);
  lds_tile_at_zero(smem);   // (round_load0 / round_store0)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const uint32_t* recs = reinterpret_cast<const uint32_t*>(coef);
  const uint32_t tile_id = blockIdx.x & ((1u << a.n_nonlocal) - 1u);
  const uint32_t pair = blockIdx.x >> a.n_nonlocal;
  const uint32_t s_a = 2u * pair, s_b = min(2u * pair + 1u, n_states - 1u);  // (an odd batch: the last state twice)
  TileCtx t = pass_tile_ctx(a, tables);
  const ThreadOff toff = thread_offsets(t, tid);
  t.tile_base = tile_base_of(a, tile_id, lane);
  const uint32_t tile_hi = tile_id << K;  // tile-bit predicates of boundary phases (cph_*): bit K + i = tile-id bit i
  float2* st_a = psi + (size_t(s_a) << a.n);
  float2* st_b = psi + (size_t(s_b) << a.n);
  const uint32_t* prog = prog_base + a.prog_off;
  uint32_t w0 = uni(prog[0]);
  if ((w0 & 0xffu) != OP_ROUND) return;  // nothing to apply (measurement-only programs never come here)
  // (the first record and the first round's thread table are requested BEFORE the tiles: one latency, not two)
  constexpr RecordLayout L(R, false);
  uint32_t pc = 0;
  uint32_t cur[1], nxt[1];
  uint32_t rec_off = uni(prog[2]);
  rec_load<1>(recs, rec_off, lane, cur);
  const uint32_t* tlt = tables + a.tl_off + uint32_t(tid);
  uint32_t DB[R], T, TL = tlt[uni(prog[3])];
  TileRegs ra, rb;
  prefetch_tile<K, NT>(ra, st_a, t, toff);
  prefetch_tile<K, NT>(rb, st_b, t, toff);
  if (a.frozen_old_local) {  // local bits nothing has acted on yet: their != input half was never written (per state)
    const uint32_t in_a = local_bits<K>(a, input_index(a, bits + size_t(state0 + s_a) * n_user, n_user, lane), lane);
    const uint32_t in_b = local_bits<K>(a, input_index(a, bits + size_t(state0 + s_b) * n_user, n_user, lane), lane);
    clear_stale<K>(ra, tid, in_a, a.frozen_old_local);
    clear_stale<K>(rb, tid, in_b, a.frozen_old_local);
  }
  round_geometry<K, R>(uni(prog[1]), TL, DB, &T);
  v2f p[NR], q[NR];
  commit_tile<K, NT>(xt, ra, tid);
  __syncthreads();
  round_load0<R>(T, DB, p);
  __syncthreads();
  commit_tile<K, NT>(xt, rb, tid);
  __syncthreads();
  round_load0<R>(T, DB, q);
  for (;;) {
    const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8;
    for (uint32_t inst = 0; inst < n_inst; ++inst) {
      rec_load<1>(recs, rec_off + L.words(), lane, nxt);  // prefetch (the buffer is padded)
      instance_fwd_pair<R, 1>(cur, recs, rec_off, p, q, TL | tile_hi);
      rec_off += L.words();
      cur[0] = nxt[0];
    }
    pc += kRoundWords;
    const uint32_t w1 = uni(prog[pc]);
    if ((w1 & 0xffu) != OP_ROUND) break;  // OP_MEASURE (ignored: see above) or OP_END
    const bool sync = !(w0 & kRoundNoBarrier);  // else the next round's waves own the same amplitudes
    uint32_t DBn[R], Tn;
    const uint32_t TLn = tlt[uni(prog[pc + 3])];
    round_geometry<K, R>(uni(prog[pc + 1]), TLn, DBn, &Tn);
    const uint32_t next_off = uni(prog[pc + 2]);
    if (next_off != rec_off) {
      rec_off = next_off;
      rec_load<1>(recs, rec_off, lane, cur);
    }
    round_store0<R>(T, DB, p);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, p);
    if (sync) __syncthreads();
    round_store0<R>(T, DB, q);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, q);
#pragma unroll
    for (int j = 0; j < R; ++j) DB[j] = DBn[j];
    T = Tn;
    TL = TLn;
    w0 = w1;
  }
  if (a.flags & PASS_STORE) {
    round_store0<R>(T, DB, p);
    __syncthreads();
    const ThreadOff o = thread_offsets(t, tid);
    store_tile<K, NT>(xt, st_a, t, o, tid);
    __syncthreads();
    round_store0<R>(T, DB, q);
    __syncthreads();
    if (s_b != s_a) store_tile<K, NT>(xt, st_b, t, o, tid);
  }
}

// ================================================================================
// Adjoint pass kernels: tile pair (psi, lambda); program already in reverse order.
// For each parametrised gate:  dE/dt = -2*pi * Im <lam| A |psi>  with psi, lam taken
// AFTER the gate and A = sum_k e_k P_k, then both are multiplied by U^dagger.
// For a diagonal term with angle t on the index set {bits all 1}:
//   dE/dt = -2*pi * sum_{selected l} Im(conj(lam_l) psi_l).
//
// Gradient partials leave a workgroup as one row of `tile_grad` [state, tile, slot of this pass]
// (no atomics anywhere: cells per wave in LDS, summed in wave order; reduce_tiles_kernel then adds
// the tiles of a state in tile order), so gradients are bit-identical from run to run and for any
// sharding of the batch over GPUs.
// ================================================================================
// One adjoint instance on the register-resident tile pair: for every parametrised micro-op the
// gradient partial Im<lam|A|psi> (reduced over the wave into its slot cell), then U^dagger on both.
template <int R, int NW, bool GEN>
__device__ __forceinline__ void instance_adj(const uint32_t (&cur)[1], const uint32_t (&sv)[1],
                                             const uint32_t* __restrict__ recs, uint32_t rec_off, int lane,
                                             uint32_t wave, v2f (&p)[1 << R], v2f (&l)[1 << R], uint32_t TLX,
                                             float* cells) {
  constexpr RecordLayout L(R, true);
  constexpr int NB = 1;
  constexpr int S0 = L.slot0();
  const RecBase rb{recs + rec_off};
  const uint32_t h0 = rec_word<0>(cur, rb), h1 = rec_word<1>(cur, rb);
  // ---- CPH (slot group 2) ----
  if (h1 & 0xffu) {
    float g[8] = {any_float(), any_float(), any_float(), any_float(), any_float(), any_float(), any_float(), any_float()};
    QHBM_FOR_RB(R,
      if ((h1 >> (2 * J)) & 1u)
        g[2 * J] = cph_adj<R, J>(p, l, rec_cs<L.cph(2 * J)>(cur, rb), rec_word<L.pred(2 * J)>(cur, rb), TLX);
      if ((h1 >> (2 * J + 1)) & 1u)
        g[2 * J + 1] = cph_adj<R, J>(p, l, rec_cs<L.cph(2 * J + 1)>(cur, rb), rec_word<L.pred(2 * J + 1)>(cur, rb), TLX);)
    add_slots8<2, NW>(cells, lane, wave, sv[0], h1 & 0xffu, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
  }
  float g1[4] = {any_float(), any_float(), any_float(), any_float()};  // PH1 partials: reduced together with the X partials (slot group 0)
  if (h1 & kFullDiagFlag) {
    // ---- all PH1/PH2 terms at once: the per-term gradients are sums of Im(conj(lam) psi) over
    // the term's index set (full_partials), then ONE conj-table multiply ----
    float g[6];
    full_partials(p, l, g1, g);
    if ((h0 >> 24) & 0x3fu) add_slots8<1, NW>(cells, lane, wave, sv[0], (h0 >> 24) & 0x3fu, g[0], g[1], g[2], g[3], g[4], g[5], any_float(), any_float());
    apply_full<NB>(p, cur, rb, false);  // (adjoint records hold the CONJUGATE phases: prep_coefs_kernel, CoefJob::dagger)
    apply_full<NB>(l, cur, rb, false);
  }
  // ---- PH2 (slot group 1) ----
  if ((h0 >> 16) & 0x3fu) {
    float g[6] = {any_float(), any_float(), any_float(), any_float(), any_float(), any_float()};
    QHBM_FOR_PAIR(R,
      if ((h0 >> (16 + pair_index(JA, JB))) & 1u) {
        const v2f cs = rec_cs<L.ph2(pair_index(JA, JB))>(cur, rb);
        g[pair_index(JA, JB)] = sum_w2<R, JA, JB>(p, l);
        apply_ph2<R, JA, JB>(p, cs);
        apply_ph2<R, JA, JB>(l, cs);
      })
    add_slots8<1, NW>(cells, lane, wave, sv[0], (h0 >> 16) & 0x3fu, g[0], g[1], g[2], g[3], g[4], g[5], any_float(), any_float());
  }
  // ---- PH1 ----
  if ((h0 >> 8) & 0xfu) {
    QHBM_FOR_RB(R,
      if ((h0 >> (8 + J)) & 1u) {
        const v2f cs = rec_cs<L.ph1(J)>(cur, rb);
        g1[J] = sum_w1<R, J>(p, l);
        apply_ph1<R, J>(p, cs);
        apply_ph1<R, J>(l, cs);
      })
  }
  // ---- one-qubit gates: X (slot group 0, with the PH1 partials), Y and dense (slot group 3) ----
  {
    float g[4] = {any_float(), any_float(), any_float(), any_float()};
    if (h0 & 0xfu) {
      QHBM_FOR_RB(R,
        if ((h0 >> J) & 1u) {
          const v2f cs = rec_cs<L.x(J)>(cur, rb);  // (U^dagger's shear coefficients: negated at preparation)
          if ((h0 >> (12 + J)) & 1u) g[J] = im_lam_x_psi<R, J>(p, l);  // (the X gates that own a gradient slot: a header bit, not a v_readlane of the slot vector)
          apply_x<R, J>(p, cs);
          apply_x<R, J>(l, cs);
        })
    }
    // the values that exist: X gates that own a slot (word 0 bits 12..15), PH1 terms (per-term: bits 8..11, FULL: 4..7)
    const uint32_t present = ((h0 >> 12) & 0xfu) | ((((h0 >> 8) | (h0 >> 4)) & 0xfu) << 4);
    if (present) add_slots8<0, NW>(cells, lane, wave, sv[0], present, g[0], g[1], g[2], g[3], g1[0], g1[1], g1[2], g1[3]);
  }
  if constexpr (GEN) {
  if ((h1 >> 16) & 0xf0fu) {
    float gy[4] = {0.f, 0.f, 0.f, 0.f}, gd[4] = {0.f, 0.f, 0.f, 0.f};
    QHBM_FOR_RB(R,
      if ((h1 >> (16 + J)) & 1u) {
        const v2f cs = rec_cs<L.y(J)>(cur, rb);
        if (rec_word<L.slot_y(J) - S0>(sv) != 0xffffffffu) gy[J] = im_lam_y_psi<R, J>(p, l);
        apply_y<R, J>(p, cs);
        apply_y<R, J>(l, cs);
      })
    if ((h1 >> 24) & 0xfu) {  // dense: U^dagger (8 floats) then generator (8 floats) per register bit
      uint32_t dv[1];
      rec_load<1>(recs, rec_off + 128u, lane, dv);
      QHBM_FOR_RB(R,
        if ((h1 >> (24 + J)) & 1u) {
          const Gen2 gen{rec_cs<16 * J + 8>(dv), rec_cs<16 * J + 10>(dv), rec_cs<16 * J + 12>(dv), rec_cs<16 * J + 14>(dv)};
          if (rec_word<L.slot_dense(J) - S0>(sv) != 0xffffffffu) gd[J] = im_lam_g1_psi<R, J>(p, l, gen);
          apply_mat1<R, J>(p, rec_cs<16 * J>(dv), rec_cs<16 * J + 2>(dv), rec_cs<16 * J + 4>(dv), rec_cs<16 * J + 6>(dv));
          apply_mat1<R, J>(l, rec_cs<16 * J>(dv), rec_cs<16 * J + 2>(dv), rec_cs<16 * J + 4>(dv), rec_cs<16 * J + 6>(dv));
        })
    }
    add_slots8<3, NW>(cells, lane, wave, sv[0], (h1 >> 16) & 0xf0fu ? (((h1 >> 16) & 0xfu) | (((h1 >> 24) & 0xfu) << 4)) : 0u, gy[0], gy[1], gy[2], gy[3], gd[0], gd[1], gd[2], gd[3]);
  }
  }
}

// Writes the tile's gradient row: slot cells summed in wave order.
template <int NT, int NW>
__device__ __forceinline__ void flush_cells(const float* cells, float* __restrict__ grow, uint32_t n_slots, int tid) {
  for (uint32_t i = tid; i < n_slots; i += NT) {
    float v = cells[i * NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += cells[i * NW + w];
    grow[i] = v;
  }
}

// The relabeling store: only the amplitudes whose newly finished bits equal the input bitstring, in
// the order of their NEW addresses (finished bits moved to the highest positions the tile owns):
// whole 128-byte lines of live data, nothing written for the dead half.
// Live out-index o -> (local index with the finished bits clear, offset in the state) is a table
// (tables[relabel_off + 2 o]) that is OR-decomposable in o: a thread looks up the entry of ITS low part of o once
// for both tiles (relabel_lookup) and ORs the entry of the iteration's high part, which the host put into the pass
// arguments -- as a loop over table entries, every iteration of both stores waited for its own table load.
struct RelabelCtx {
  uint32_t l_mine, off_mine;  // the thread's own entry
  uint32_t fz_addr;           // where the finished bits go in the address, carrying the input bits
  uint32_t fz_local;          // the finished bits of the live amplitudes, local index space
  uint32_t count;             // out-indices (pairs of them when relabel_pairs) the store covers
};
template <int K>
__device__ __forceinline__ RelabelCtx relabel_lookup(const PassArgs& a, const uint32_t* __restrict__ tables, uint32_t in_local,
                                                     int tid, int lane) {
  RelabelCtx r;
  const uint32_t n_live = uint32_t(K) - a.n_fz;
  r.count = a.relabel_pairs ? 1u << (n_live - 1u) : 1u << n_live;
  const uint32_t mine = min(uint32_t(tid), r.count - 1u) << (a.relabel_pairs ? 1 : 0);
  const uint2 e = *reinterpret_cast<const uint2*>(tables + a.relabel_off + 2u * mine);
  const uint32_t src = a.fz_src[uint32_t(lane) & 31u];
  r.fz_addr = ballot32((lane < 32) & (src != 0xffu) & (((in_local >> (src & 31u)) & 1u) != 0u));
  r.fz_local = in_local & a.frozen_new_local;
  r.l_mine = e.x;
  r.off_mine = e.y;
  return r;
}
template <int K, int NT>
__device__ __forceinline__ void store_tile_relabeled(const float2* __restrict__ tile, float2* __restrict__ st,
                                                     const PassArgs& a, const RelabelCtx& r, uint32_t tile_base, int tid) {
  float2* sb = st + (tile_base | r.fz_addr);
  if (a.relabel_pairs) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (uint32_t(i) < a.relabel_iters && uint32_t(tid) + uint32_t(i * NT) < r.count) {
        const uint32_t l0 = r.l_mine | a.relabel_hi[i][0] | r.fz_local;
        const float2 v0 = tile[swz(l0)], v1 = tile[swz(l0 | a.relabel_l1)];
        *reinterpret_cast<float4*>(sb + (r.off_mine | a.relabel_hi[i][1])) = make_float4(v0.x, v0.y, v1.x, v1.y);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (uint32_t(i) < a.relabel_iters && uint32_t(tid) + uint32_t(i * NT) < r.count)
        sb[r.off_mine | a.relabel_hi[i][1]] = tile[swz(r.l_mine | a.relabel_hi[i][0] | r.fz_local)];
    }
  }
}

#ifdef QHBM_ADJ_TIMING  // diagnostic build (scripts/experiments/ablate/build.py adj_timing; never shipped): the cycles a
// wave of the exchange adjoint kernel spends per phase, summed over the waves of a launch, printed by the launcher
__device__ unsigned long long g_adj_phase[8 * 64];
#define QHBM_TICK() __builtin_readcyclecounter()
#define QHBM_PHASE(k, v) if (lane == 0) atomicAdd(&g_adj_phase[(k) * 64 + (blockIdx.x & 63u)], (unsigned long long)(v))
#else
#define QHBM_TICK() 0ull
#define QHBM_PHASE(k, v)
#endif

// ---- exchange layout (default for passes without Y / dense gates) ---------------------------------
// The (psi, lambda) tile pair lives in REGISTERS for the whole pass; LDS holds one tile-sized
// exchange buffer through which psi, then lambda, change geometry between rounds.  Half the LDS
// of the two-tile layout: four waves per SIMD instead of two (K = 12: 36 KiB per workgroup, four
// workgroups per CU).  A round boundary whose waves keep their amplitudes needs no barrier at all;
// otherwise three (store psi | load psi | store lambda | load lambda): a wave always writes the
// region it last read, so nothing else can race.
template <int K, int ROWS>
__global__ __launch_bounds__(1 << (K - 4), adjx_min_waves(K)) void pass_adjx_kernel(
    PassArgs a, float2* __restrict__ psi, float2* __restrict__ lam, const int8_t* __restrict__ bits, int n_user,
    const uint32_t* __restrict__ prog_base, const uint32_t* __restrict__ tables,
    const float* __restrict__ coef, float* __restrict__ tile_grad /*[states * tiles, n_slots]*/, uint32_t state0) {
  constexpr int R = 4;
  constexpr int NT = 1 << (K - R);
  constexpr int NR = 1 << R;
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* xt = reinterpret_cast<float2*>(smem);
  float* cells = reinterpret_cast<float*>(xt + (1 << K));  // [kMaxSlotsPerPass][NW]
  lds_tile_at_zero(smem);   // (round_load0 / round_store0)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const uint32_t wave = uni(uint32_t(tid) >> 6);
  const uint32_t* recs = reinterpret_cast<const uint32_t*>(coef);
  const uint32_t s_local = blockIdx.x >> a.n_free;
  [[maybe_unused]] const unsigned long long tk0 = QHBM_TICK();  // (QHBM_ADJ_TIMING builds only: 0 otherwise)
  [[maybe_unused]] unsigned long long tk_inst = 0, tk_xchg = 0;
  // (the input bitstring in the layout this pass loads: relabeling plans move finished bits)
  TileCtx t = pass_tile_ctx(a, tables);
  const ThreadOff toff = thread_offsets<ROWS>(t, tid);
  const uint32_t idx = input_index(a, bits + size_t(state0 + s_local) * n_user, n_user, lane);
  // tail of the sweep: the tiles on which psi is identically zero are not launched (launched_tile)
  const uint32_t tile_id = launched_tile(a, blockIdx.x, idx, lane);
  t.tile_base = tile_base_of(a, tile_id, lane);
  const uint32_t tile_hi = tile_id << K;  // tile-bit predicates of boundary phases (cph_*): bit K + i = tile-id bit i
  float* grow = tile_grad + size_t(blockIdx.x) * a.n_slots;
  const uint32_t* prog = prog_base + a.prog_off;
  uint32_t w0 = uni(prog[0]);
  const bool skip = (w0 & 0xffu) != OP_ROUND;  // (an empty program: nothing to un-apply)
  const uint32_t in_local = local_bits<K>(a, idx, lane);  // the input bitstring on the tile's local bits (OP_ROUND word 4: dead waves)
  if (skip) {
    for (uint32_t i = tid; i < a.n_slots; i += NT) grow[i] = 0.f;
    return;
  }
  float2* sp = psi + (size_t(s_local) << a.n);
  float2* sl = lam + (size_t(s_local) << a.n);
  // (the first record, its slot words and the first round's thread table are requested BEFORE the tiles: one latency)
  constexpr RecordLayout L(R, true);
  uint32_t pc = 0;
  uint32_t cur[1], nxt[1], sv[1], svn[1];
  uint32_t rec_off = uni(prog[2]);
  rec_load<1>(recs, rec_off, lane, cur);
  rec_load<1>(recs, rec_off + L.slot0(), lane, sv);
  const uint32_t* tlt = tables + a.tl_off + uint32_t(tid);
  uint32_t DB[R], T, TL = tlt[uni(prog[3])];
  TileRegs rp, rl;
  prefetch_tile<K, NT, ROWS>(rp, sp, t, toff);
  prefetch_tile<K, NT, ROWS>(rl, sl, t, toff);
  [[maybe_unused]] const unsigned long long tk1 = QHBM_TICK();
  if (a.frozen_old_local) {
    clear_stale<K>(rp, tid, in_local, a.frozen_old_local);
    clear_stale<K>(rl, tid, in_local, a.frozen_old_local);
  }
  for (uint32_t i = tid; i < a.n_slots * NW; i += NT) cells[i] = 0.f;
  round_geometry<K, R>(uni(prog[1]), TL, DB, &T);
  v2f p[NR], l[NR];
  commit_tile<K, NT>(xt, rp, tid);
  __syncthreads();
  round_load0<R>(T, DB, p);
  __syncthreads();
  commit_tile<K, NT>(xt, rl, tid);
  __syncthreads();
  round_load0<R>(T, DB, l);
  [[maybe_unused]] const unsigned long long tk2 = QHBM_TICK();
  [[maybe_unused]] unsigned long long tk_r = tk2;
  for (;;) {
    const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8;
    // A wave whose (wave-index) bits differ from the input bitstring on a bit that no non-diagonal gate
    // will touch any more holds zeros of psi: nothing it could add to a gradient, nothing a later
    // round reads from it but zeros -- it skips the instances and only takes part in the exchange.
    const bool dead = uni((TL ^ in_local) & uni(prog[pc + 4])) != 0;
    if (dead) {
      rec_off += n_inst * L.words();
      rec_load<1>(recs, rec_off, lane, cur);
      rec_load<1>(recs, rec_off + L.slot0(), lane, sv);
    } else {
      for (uint32_t inst = 0; inst < n_inst; ++inst) {
        rec_load<1>(recs, rec_off + L.words(), lane, nxt);  // prefetch (the buffer is padded)
        rec_load<1>(recs, rec_off + L.words() + L.slot0(), lane, svn);
        instance_adj<R, NW, false>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);
        rec_off += L.words();
        cur[0] = nxt[0];
        sv[0] = svn[0];
      }
    }
    pc += kRoundWords;
    { [[maybe_unused]] const unsigned long long now = QHBM_TICK(); tk_inst += now - tk_r; tk_r = now; }
    const uint32_t w1 = uni(prog[pc]);
    if ((w1 & 0xffu) != OP_ROUND) break;
    // ---- change of geometry through the exchange buffer ----
    const bool sync = !(w0 & kRoundNoBarrier);  // else the next round's waves own the same amplitudes
    uint32_t DBn[R], Tn;
    const uint32_t TLn = tlt[uni(prog[pc + 3])];
    round_geometry<K, R>(uni(prog[pc + 1]), TLn, DBn, &Tn);
    const uint32_t next_off = uni(prog[pc + 2]);
    if (next_off != rec_off) {  // records of consecutive rounds are consecutive: normally already prefetched
      rec_off = next_off;
      rec_load<1>(recs, rec_off, lane, cur);
      rec_load<1>(recs, rec_off + L.slot0(), lane, sv);
    }
    round_store0<R>(T, DB, p);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, p);
    if (sync) __syncthreads();
    round_store0<R>(T, DB, l);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, l);
#pragma unroll
    for (int j = 0; j < R; ++j) DB[j] = DBn[j];
    T = Tn;
    TL = TLn;
    w0 = w1;
    { [[maybe_unused]] const unsigned long long now = QHBM_TICK(); tk_xchg += now - tk_r; tk_r = now; }
  }
  // (sched_barrier: the epilogue runs once per workgroup -- nothing to gain from interleaving the two tiles' stores, and
  // hoisting the second tile's LDS reads over the first tile's stores spilled 4 - 13 VGPRs in the ROWS8 / K = 10, 11, 13
  // instantiations)
  if (a.flags & PASS_RELABEL) {
    const RelabelCtx rc = relabel_lookup<K>(a, tables, in_local, tid, lane);  // (in flight under the exchange below)
    round_store0<R>(T, DB, p);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_tile_relabeled<K, NT>(xt, sp, a, rc, t.tile_base, tid);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    round_store0<R>(T, DB, l);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_tile_relabeled<K, NT>(xt, sl, a, rc, t.tile_base, tid);
  } else if (a.flags & PASS_STORE) {
    const ThreadOff o = thread_offsets<ROWS>(t, tid);
    round_store0<R>(T, DB, p);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_tile<K, NT, ROWS>(xt, sp, t, o, tid);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    round_store0<R>(T, DB, l);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    store_tile<K, NT, ROWS>(xt, sl, t, o, tid);
  } else {
    __syncthreads();  // the cells of every wave are complete
  }
  flush_cells<NT, NW>(cells, grow, a.n_slots, tid);
#ifdef QHBM_ADJ_TIMING
  {
    const unsigned long long tk4 = QHBM_TICK();
    QHBM_PHASE(0, tk1 - tk0);   // entry -> both tiles requested
    QHBM_PHASE(1, tk2 - tk1);   // tiles arrive, staged through LDS into the first round's geometry
    QHBM_PHASE(2, tk_inst);     // instances
    QHBM_PHASE(3, tk_xchg);     // changes of geometry between rounds
    QHBM_PHASE(4, tk4 - tk_r);  // stores, gradient row
    QHBM_PHASE(5, tk4 - tk0);   // the wave's life
    QHBM_PHASE(6, 1);
  }
#endif
}

// ---- two-tile layout: both tiles resident in LDS (programs with Y / dense 2x2 / dense two-qubit
// ops, which work on the LDS tiles directly; also selectable for A/B measurements) -------------
template <int K, bool GEN>
__global__ __launch_bounds__(1 << (K - 4), adj_min_waves(K)) void pass_adj_kernel(
    PassArgs a, float2* __restrict__ psi, float2* __restrict__ lam, const int8_t* __restrict__ bits, int n_user,
    const uint32_t* __restrict__ prog_base, const uint32_t* __restrict__ tables,
    const float* __restrict__ coef, float* __restrict__ tile_grad /*[states * tiles, n_slots]*/, uint32_t state0) {
  constexpr int R = 4;
  constexpr int NT = 1 << (K - R);
  constexpr int NR = 1 << R;
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tp = reinterpret_cast<float2*>(smem);
  float2* tl = tp + (1 << K);
  float* cells = reinterpret_cast<float*>(tl + (1 << K));  // [kMaxSlotsPerPass][NW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const uint32_t wave = uni(uint32_t(tid) >> 6);
  const uint32_t* recs = reinterpret_cast<const uint32_t*>(coef);
  const uint32_t s_local = blockIdx.x >> a.n_free;
  // tail of the sweep: the tiles on which psi is identically zero are not launched (launched_tile; these plans
  // never relabel: logical = physical index bits)
  const uint32_t tile_id = launched_tile(a, blockIdx.x, input_index(a, bits + size_t(state0 + s_local) * n_user, n_user, lane), lane);
  TileCtx t = pass_tile_ctx(a, tables);
  t.tile_base = tile_base_of(a, tile_id, lane);
  const uint32_t tile_hi = tile_id << K;  // tile-bit predicates of boundary phases (cph_*): bit K + i = tile-id bit i
  float* grow = tile_grad + size_t(blockIdx.x) * a.n_slots;
  float2* sp = psi + (size_t(s_local) << a.n);
  float2* sl = lam + (size_t(s_local) << a.n);
  {
    TileRegs rp, rl;
    const ThreadOff o = thread_offsets<ROWS_TEST>(t, tid);
    prefetch_tile<K, NT, ROWS_TEST>(rp, sp, t, o);
    prefetch_tile<K, NT, ROWS_TEST>(rl, sl, t, o);
    commit_tile<K, NT>(tp, rp, tid);
    commit_tile<K, NT>(tl, rl, tid);
  }
  for (uint32_t i = tid; i < a.n_slots * NW; i += NT) cells[i] = 0.f;
  __syncthreads();

  const uint32_t* prog = prog_base + a.prog_off;
  uint32_t pc = 0;
  uint32_t cur[1], nxt[1], sv[1], svn[1];
  uint32_t carried_off = 0xffffffffu;  // record offset whose words `cur` / `sv` hold
  for (;;) {
    const uint32_t w0 = uni(prog[pc]);
    const uint32_t opc = w0 & 0xffu;
    if (opc == OP_END) break;
    if (opc == OP_ROUND) {
      const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8;
      const uint32_t regmask = uni(prog[pc + 1]);
      uint32_t rec_off = uni(prog[pc + 2]);
      constexpr RecordLayout L(R, true);
      if (rec_off != carried_off) {  // else: prefetched by the previous round's last instance
        rec_load<1>(recs, rec_off, lane, cur);
        rec_load<1>(recs, rec_off + L.slot0(), lane, sv);
      }
      uint32_t DB[R], T;
      const uint32_t TL = tables[a.tl_off + uni(prog[pc + 3]) + uint32_t(tid)];
      round_geometry<K, R>(regmask, TL, DB, &T);
      v2f p[NR], l[NR];
      round_load<R>(tp, T, DB, p);
      round_load<R>(tl, T, DB, l);
      for (uint32_t inst = 0; inst < n_inst; ++inst) {
        rec_load<1>(recs, rec_off + L.words(), lane, nxt);  // prefetch (the buffer is padded)
        rec_load<1>(recs, rec_off + L.words() + L.slot0(), lane, svn);
        instance_adj<R, NW, GEN>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);
        rec_off += L.words();
        cur[0] = nxt[0];
        sv[0] = svn[0];
      }
      carried_off = rec_off;
      round_store<R>(tp, T, DB, p);
      round_store<R>(tl, T, DB, l);
      if (!(w0 & kRoundNoBarrier)) __syncthreads();  // else the next round's waves read only their own writes
      pc += kRoundWords;
    } else {  // OP_GATE2
      if constexpr (GEN) {
      const uint32_t pw = uni(prog[pc + 1]);
      const uint32_t pos0 = pw & 0xffu, pos1 = (pw >> 8) & 0xffu;
      const float* cf = coef + uni(prog[pc + 2]);  // U^dagger (32 floats) then generator (32 floats)
      const uint32_t slot = uni(prog[pc + 3]);
      float gacc = 0.f;
      for (uint32_t q = tid; q < (1u << (K - 2)); q += NT) {
        uint32_t ix[4];
        quad_indices(q, pos0, pos1, ix);
        float2 x[4], l[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = tp[ix[j]]; l[j] = tl[ix[j]]; }
        if (slot != 0xffffffffu) {
          mat4_apply(cf + 32, x, y);  // G psi
#pragma unroll
          for (int j = 0; j < 4; ++j) gacc += l[j].x * y[j].y - l[j].y * y[j].x;  // Im(conj(lam) * G psi)
        }
        mat4_apply(cf, x, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) tp[ix[j]] = y[j];
        mat4_apply(cf, l, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) tl[ix[j]] = y[j];
      }
      if (slot != 0xffffffffu) add_slot<NW>(cells, tid, slot, gacc);
      __syncthreads();
      }
      pc += kGate2Words;
    }
  }
  __syncthreads();
  flush_cells<NT, NW>(cells, grow, a.n_slots, tid);
  if (a.flags & PASS_STORE) {
    const ThreadOff o = thread_offsets<ROWS_TEST>(t, tid);
    store_tile<K, NT, ROWS_TEST>(tp, sp, t, o, tid);
    store_tile<K, NT, ROWS_TEST>(tl, sl, t, o, tid);
  }
}

// Sum of a state's tile_grad rows in a FIXED tree (bit-reproducible): a block adds a chunk of up to
// kTileChunk consecutive tiles -- lane y takes tiles y, y + 16, ... in order, then the 16 partials are
// added in lane order -- and writes row `chunk` of dst[s]; the host repeats the step on the chunk
// sums until one row per state is left, which lands in state_grad.
constexpr uint32_t kTileChunk = 256;
__global__ __launch_bounds__(1024) void reduce_tiles_kernel(const float* __restrict__ src, uint32_t n_rows,
                                                            uint32_t n_slots, float* __restrict__ dst,
                                                            uint32_t dst_row_stride, uint32_t dst_state_stride,
                                                            uint32_t dst_offset) {
  __shared__ float part[16][64];
  const uint32_t s = blockIdx.z, chunk = blockIdx.y;
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  const uint32_t r0 = chunk * kTileChunk, r1 = min(n_rows, r0 + kTileChunk);
  const float* base = src + size_t(s) * n_rows * n_slots;
  float acc = 0.f;
  if (i < n_slots)
    for (uint32_t r = r0 + threadIdx.y; r < r1; r += 16) acc += base[size_t(r) * n_slots + i];
  part[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && i < n_slots) {
    float v = part[0][threadIdx.x];
#pragma unroll
    for (int y = 1; y < 16; ++y) v += part[y][threadIdx.x];
    dst[size_t(s) * dst_state_stride + size_t(chunk) * dst_row_stride + dst_offset + i] = v;
  }
}

// ================================================================================
// lambda = sum_k upstream[s, op_k] * c_k * P_k psi,   (P psi)[j] = i^ny (-1)^{popc((j^x) & z)} psi[j ^ x]
// Terms arrive sorted by x mask and cut into GROUPS of equal x: one gather psi[j ^ x] per group
// serves all of its terms (XXZ: 57 terms, 20 gathers), whose signed weights -- staged once per
// workgroup in LDS as upstream * coeff * i^ny -- fold into one complex factor per amplitude.
//
// A workgroup owns a block of 256 A consecutive amplitudes; a thread owns A of them as A / 2 ADJACENT
// PAIRS, local index L(s) = 2 tid + (s & 1) + 512 (s >> 1) for slot s: the partner j ^ x of an adjacent
// pair is an adjacent pair, so every gather is ONE 16-byte load per pair (half the load instructions
// of 8-byte gathers, twice the bytes in flight per instruction).  The kernel is bound by the LATENCY of
// those gathers (round 3: 58 % of wave-cycles in s_waitcnt, HBM at 0.57 of its rate, the fill traffic
// cut by a quarter without any effect on the time), so the loads of the NEXT group that leaves the
// block are issued before the current group is consumed, and the staging area is small enough for
// five workgroups per CU.  Masks inside the block read the block's own amplitudes from LDS.
// The sign of a term at j ^ x is (-1)^{popc((j ^ x) & z)}: the part from the block and slot bits of j
// and from x & z is the same for the whole workgroup, so it is folded into A pre-signed copies of the
// weight when the term is staged; a thread evaluates only the parity of its thread bits & z, once per
// term for all of its amplitudes.  The terms of a group come sorted by sign class (kernels.h
// ObsGroup): the ones whose z misses the thread bits are pre-summed per slot once per workgroup, the
// ones whose z misses the slot bits cost one signed sum per THREAD, only the rest are added per
// amplitude (XXZ at 20 qubits: 2 of the 57 terms), and lambda accumulates with one packed FMA per
// amplitude and group.
// ================================================================================
template <int A> struct ObsStage {
  float4 term[kObsTermChunk / 2];     // z bits, weight (re), weight (im), --      (general path)
  float flip[kObsTermChunk / 2][A];   // real weight, pre-signed for slot s         (real-weight path)
  uint32_t live[kObsTermChunk / 2];   // [first term of a group] pairs of slots whose weight is not exactly zero
};
constexpr uint32_t kObsChunk = kObsTermChunk / 2;  // terms staged in LDS at a time
// index bits of slot s inside the block (kernels.h obs_slot_mask is their union)
template <int A> __device__ __forceinline__ constexpr uint32_t obs_slot_bits(int s) { return uint32_t(s & 1) | (uint32_t(s >> 1) << 9); }
// the partners j ^ x of a thread's A / 2 adjacent pairs: one 16-byte load per pair (x with bit 0 cleared)
// Only the pairs of `live` are fetched: the terms of a mask often cancel on part of the index space -- XX + YY on a
// pair of qubits is zero where the two bits are equal --, and where the mask's qubits are block or slot bits the
// whole workgroup sees weight 0 for those slots (config 3: half of the partner runs).  A pair that is not
// fetched holds the thread's own amplitudes instead and is multiplied by exactly 0.
// FAR launches (two-level lambda = O psi at >= 26 qubits, engine.cpp far_windows): the kernel works on a VIRTUAL index in
// which seven far index bits [far_hi, far_hi + 7) have changed places with bits [4, 11) -- the masks that flip only those
// far bits then permute a workgroup's own block (served from its LDS copy, no gather) while its amplitudes still come
// as 128-byte runs (bits 0..3 stay).  obs_phys: virtual -> physical index.
template <bool FAR>
__device__ __forceinline__ uint32_t obs_phys(uint32_t v, uint32_t far_hi) {
  if constexpr (!FAR) return v;
  const uint32_t lo = (v >> 4) & 0x7fu, hi = (v >> far_hi) & 0x7fu;
  return (v & ~((0x7fu << 4) | (0x7fu << far_hi))) | (hi << 4) | (lo << far_hi);
}
template <int A, bool FAR = false>
__device__ __forceinline__ void obs_gather(float4 (&buf)[A / 2], const float2* __restrict__ ps, uint32_t j0, uint32_t x,
                                           uint32_t live, uint32_t far_hi = 0) {
  if (live == 0) return;
#pragma unroll
  for (int p = 0; p < A / 2; ++p)  // (no branch per pair: a dead one re-reads the thread's own pair, a hit in the nearest cache)
    buf[p] = *reinterpret_cast<const float4*>(&ps[obs_phys<FAR>((j0 + 512u * p) ^ ((live >> p) & 1u ? x & ~1u : 0u), far_hi)]);
}
// One group: acc[s] += (signed weight sum of the group's terms at slot s) * psi[j ^ x].  `buf` holds the
// gathered partners if the mask leaves the block; a mask inside the block reads the block's LDS copy.
// MULTI (several observables, OBS_GATHER_MULTI): the staged weights are the bare coefficients of ONE observable (the
// group's), `upw` its upstream weight for lambda (0 when no lambda is formed), and <psi|O_op|psi> collects
// sum_a Re(conj(psi_a) * (weight_a * psi[a ^ x])) in `eval`.
template <int A, bool MULTI = false>
__device__ __forceinline__ void obs_consume(const ObsGroup& gr, float4 (&buf)[A / 2], const v2f* own, const ObsStage<A>& st,
                                            uint32_t& k, uint32_t k0, uint32_t tb, v2f (&acc)[A], float upw = 1.f,
                                            float* eval = nullptr) {
  constexpr int P = A / 2;
  if (gr.x < 256u * A) {  // the mask permutes the workgroup's own block: served from its LDS copy
#pragma unroll
    for (int p = 0; p < P; ++p) buf[p] = *reinterpret_cast<const float4*>(&own[(tb + 512u * p) ^ (gr.x & ~1u)]);
  }
  // index bit 0 flipped (odd): the two amplitudes of a pair change places -- the accumulation is written
  // out for both orders under a scalar branch (a select per register would cost sixteen v_cndmask a group)
  const bool odd = uni(gr.x) & 1u;
  if (gr.has_imag) {
    float cr[A], ci[A];
#pragma unroll
    for (int a = 0; a < A; ++a) cr[a] = ci[a] = 0.f;
    for (; k < gr.end; ++k) {
      const float4 t = st.term[k - k0];
      const uint32_t z = __float_as_uint(t.x);
      const uint32_t sg0 = (uint32_t(__popc(tb & z)) << 31) ^ __float_as_uint(t.w);
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const uint32_t sgn = sg0 ^ (uint32_t(__popc(obs_slot_bits<A>(a) & z)) << 31);
        cr[a] += __uint_as_float(__float_as_uint(t.y) ^ sgn);
        ci[a] += __uint_as_float(__float_as_uint(t.z) ^ sgn);
      }
    }
    // (both orders of a pair written out under the scalar branch, as in the real-weight path: `odd ? hi : lo` per
    // register compiles to sixteen v_cndmask_b32 on VCC, 23 cycles each on gfx950)
    auto weigh = [&](int p, v2f v0, v2f v1) {
      const v2f t0 = v2f{cr[2 * p] * v0.x - ci[2 * p] * v0.y, cr[2 * p] * v0.y + ci[2 * p] * v0.x};
      const v2f t1 = v2f{cr[2 * p + 1] * v1.x - ci[2 * p + 1] * v1.y, cr[2 * p + 1] * v1.y + ci[2 * p + 1] * v1.x};
      if constexpr (MULTI) {
        const float4 mine = *reinterpret_cast<const float4*>(&own[tb + 512u * p]);
        *eval += (mine.x * t0.x + mine.y * t0.y) + (mine.z * t1.x + mine.w * t1.y);
        acc[2 * p] += upw * t0;
        acc[2 * p + 1] += upw * t1;
      } else {
        acc[2 * p] += t0;
        acc[2 * p + 1] += t1;
      }
    };
    if (odd) {  // (the asm comments keep the optimiser from folding the two branches back into selects)
      asm volatile("; imaginary weights, pairs swapped");
#pragma unroll
      for (int p = 0; p < P; ++p) weigh(p, v2f{buf[p].z, buf[p].w}, v2f{buf[p].x, buf[p].y});
    } else {
      asm volatile("; imaginary weights, pairs in place");
#pragma unroll
      for (int p = 0; p < P; ++p) weigh(p, v2f{buf[p].x, buf[p].y}, v2f{buf[p].z, buf[p].w});
    }
  } else {  // real weights only (X/Z strings, even Y count): the common case
    float c[A];
    float sgl = 0.f;  // terms whose sign depends on the thread only: one signed sum for all A amplitudes
    const uint32_t kl = k + gr.n_h, km = kl + gr.n_l;
    for (uint32_t q = kl; q < km; ++q) {
      const uint32_t z = __float_as_uint(st.term[q - k0].x);
      sgl += __uint_as_float(__float_as_uint(st.flip[q - k0][0]) ^ (uint32_t(__popc(tb & z)) << 31));
    }
    {  // pre-summed weights of the thread-independent terms: ONE vector read of the slot row, no branch per
      // slot (eight conditional 4-byte reads, each with its own wait, were a chain of LDS latencies per group)
      const float use_h = gr.n_h ? 1.f : 0.f;  // wave-uniform; the row read is valid memory either way
#pragma unroll
      for (int q4 = 0; q4 < A / 4; ++q4) {
        const float4 r4 = *reinterpret_cast<const float4*>(&st.flip[k - k0][4 * q4]);
        c[4 * q4] = fmaf(use_h, r4.x, sgl);
        c[4 * q4 + 1] = fmaf(use_h, r4.y, sgl);
        c[4 * q4 + 2] = fmaf(use_h, r4.z, sgl);
        c[4 * q4 + 3] = fmaf(use_h, r4.w, sgl);
      }
    }
    for (k = km; k < gr.end; ++k) {
      const uint32_t z = __float_as_uint(st.term[k - k0].x);
      const uint32_t sg0 = uint32_t(__popc(tb & z)) << 31;
#pragma unroll
      for (int a = 0; a < A; ++a) c[a] += __uint_as_float(__float_as_uint(st.flip[k - k0][a]) ^ sg0);
    }
    if constexpr (MULTI) {  // the value first (bare coefficients), then lambda with the observable's upstream weight
      float e = 0.f;
      if (odd) {  // (slot 2p pairs with the partner pair's second amplitude, 2p + 1 with its first)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float4 mine = *reinterpret_cast<const float4*>(&own[tb + 512u * p]);
          e += c[2 * p] * (mine.x * buf[p].z + mine.y * buf[p].w) + c[2 * p + 1] * (mine.z * buf[p].x + mine.w * buf[p].y);
        }
      } else {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float4 mine = *reinterpret_cast<const float4*>(&own[tb + 512u * p]);
          e += c[2 * p] * (mine.x * buf[p].x + mine.y * buf[p].y) + c[2 * p + 1] * (mine.z * buf[p].z + mine.w * buf[p].w);
        }
      }
      *eval += e;
#pragma unroll
      for (int a = 0; a < A; ++a) c[a] *= upw;
    }
    if (odd) {
#pragma unroll
      for (int p = 0; p < P; ++p) {  // one packed FMA per amplitude
        acc[2 * p] += c[2 * p] * v2f{buf[p].z, buf[p].w};
        acc[2 * p + 1] += c[2 * p + 1] * v2f{buf[p].x, buf[p].y};
      }
    } else {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        acc[2 * p] += c[2 * p] * v2f{buf[p].x, buf[p].y};
        acc[2 * p + 1] += c[2 * p + 1] * v2f{buf[p].z, buf[p].w};
      }
    }
  }
}
// VALUE (a single observable): the weights are the bare coefficients, lambda = O psi unweighted, and
// <psi|O|psi> = sum_j Re(conj(psi_j) lambda_j) leaves as a by-product in the fixed-point accumulator
// value_part[state, workgroup] (value_parts_kernel) -- the forward sweep then needs no measurement at all, and the caller applies the
// upstream weight to the state's gradient row (the adjoint sweep is linear in lambda).
// VM = OBS_GATHER_MULTI: 2..4 observables, one launch for the weighted lambda AND every <psi|O_t|psi> (round 4 took
// the values from a second launch of the block kernel: config 3 as XX / YY / ZZ sums 64 ms for the two instead of 24).
// The terms come sorted by (x, observable); a group is one observable's share of a mask, carries the BARE coefficients
// (the upstream weight of its observable multiplies the group's folded weights when lambda is formed), and a group on
// the mask of its predecessor re-uses the gathered partners.  No partner run can be skipped as vanishing: XX + YY
// cancels where the two bits agree, XX alone and YY alone do not.
// FAR = true: a far launch of the two-level sweep (obs_phys): virtual indices, lambda and the value partials ACCUMULATED
// onto what the first launch left.
template <int A, int VM, bool FAR = false>
__global__ __launch_bounds__(256, VM == OBS_GATHER_MULTI ? 4 : 5) void apply_observable_kernel(
    const float2* __restrict__ psi, float2* __restrict__ lam, uint32_t n, const DevTerm* __restrict__ terms,
    uint32_t n_terms, const ObsGroup* __restrict__ groups, uint32_t n_groups,
    const float* __restrict__ upstream, uint32_t n_ops, uint32_t state0, float* __restrict__ value_part,
    uint32_t nb /* workgroups per state */, uint32_t n_states, uint32_t xcd_states, uint32_t far_hi) {
  constexpr int P = A / 2;  // adjacent pairs per thread
  constexpr bool VALUE = VM == OBS_GATHER_VALUE, MULTI = VM == OBS_GATHER_MULTI;
  __shared__ ObsStage<A> st;
  // Workgroups are dealt round-robin to the 8 XCDs (linear id mod 8), each with its own L2.
  //   xcd_states: XCD k works on state 8 g + k, its blocks in index order -- EVERY partner run j ^ x of
  //   that state is fetched into the same L2, by this workgroup or by the one that owns it (config 3:
  //   3.7 reads of the state from the fabric instead of 5.1).  The last n_states mod 8 states, and tiny
  //   states, use the other map:
  //   XCD k takes the k-th CONTIGUOUS eighth of every state, so that the partners of every mask below
  //   that eighth's size share an L2; the masks above it are fetched from another XCD's share.
  uint32_t s_local, bx;
  {
    const uint32_t wg = blockIdx.x, per_group = 8u * nb, group = wg / per_group, r = wg - group * per_group;
    if (xcd_states && (group + 1u) * 8u <= n_states) {
      s_local = group * 8u + (r & 7u);
      bx = r >> 3;
    } else {
      const uint32_t q = r / nb, b = r - q * nb;
      s_local = group * 8u + q;
      bx = (nb & 7u) ? b : (b & 7u) * (nb >> 3) + (b >> 3);
    }
  }
  const uint32_t jb = bx * (256u * A);                   // block bits of j
  const uint32_t tb = threadIdx.x << 1;                  // thread bits of j
  const float2* ps = psi + (size_t(s_local) << n);
  const uint32_t j0 = jb + tb;                           // the thread's first amplitude
  const float* up = (VALUE || !upstream) ? nullptr : upstream + size_t(state0 + s_local) * n_ops;
  __shared__ __attribute__((aligned(16))) v2f own[256 * A];
  float4 self[P];
#pragma unroll
  for (int p = 0; p < P; ++p) self[p] = *reinterpret_cast<const float4*>(&ps[obs_phys<FAR>(j0 + 512u * p, far_hi)]);
  // gathers of the first group that leaves the block (if the first group does): in flight during the staging
  ObsGroup gr = n_groups ? groups[0] : ObsGroup{0u, 0u, 0u, 0u, 0u};
  float4 cur[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    *reinterpret_cast<float4*>(&own[tb + 512u * p]) = self[p];
  }
  v2f acc[A];
#pragma unroll
  for (int a = 0; a < A; ++a) acc[a] = v2f{0.f, 0.f};
  float ev[4] = {0.f, 0.f, 0.f, 0.f};  // MULTI: <psi|O_t|psi> of this thread's amplitudes
  uint32_t g = 0;
  for (uint32_t k0 = 0; k0 < n_terms; k0 += kObsChunk) {
    const uint32_t k1 = min(n_terms, k0 + kObsChunk);
    if (k0) __syncthreads();
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256u) {
      const DevTerm tm = terms[k];
      const float w = (VALUE || MULTI) ? tm.coeff : up[tm.op] * tm.coeff;
      const float m = (tm.ny & 2u) ? -w : w;
      // parity of the workgroup-constant part: block bits of j and the x & z overlap (= ny)
      const uint32_t base = (uint32_t(__popc(jb & tm.z)) + tm.ny) & 1u;
      st.term[k - k0] = make_float4(__uint_as_float(tm.z), (tm.ny & 1u) ? 0.f : m, (tm.ny & 1u) ? m : 0.f,
                                    __uint_as_float(base << 31));
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const uint32_t par = (base + uint32_t(__popc(obs_slot_bits<A>(a) & tm.z))) & 1u;
        st.flip[k - k0][a] = __uint_as_float(__float_as_uint(m) ^ (par << 31));
      }
    }
    __syncthreads();
    // one pre-summed weight per slot for the thread-independent terms of every real group
    // of the chunk, written over the first of them (flip[begin][a])
    // ... and the pairs of slots the group does not vanish on (obs_gather): all of them unless every term of
    // the group is thread-independent and the weights of both slots of a pair add up to exactly 0
    for (uint32_t gi = g + threadIdx.x / uint32_t(A); gi < n_groups; gi += 256u / uint32_t(A)) {
      const ObsGroup gq = groups[gi];
      if (gq.end > k1) break;
      const uint32_t kb = gi ? groups[gi - 1u].end : 0u, a = threadIdx.x % uint32_t(A);
      float sum = st.flip[kb - k0][a];
      if (gq.n_h > 1u) {
        for (uint32_t k = kb + 1u; k < kb + gq.n_h; ++k) sum += st.flip[k - k0][a];
        st.flip[kb - k0][a] = sum;
      }
      // the A lanes of a group are consecutive lanes of one wave
      const uint32_t nz = uint32_t(__ballot(sum != 0.f) >> ((threadIdx.x & 63u) / uint32_t(A) * uint32_t(A))) & ((1u << A) - 1u);
      uint32_t pairs = 0;
#pragma unroll
      for (int p = 0; p < P; ++p) pairs |= ((nz >> (2 * p)) & 3u) ? 1u << p : 0u;
      // ((mask, observable) order -- same_x bit 1 --: a group may feed on its predecessor's partners: every pair is fetched)
      if (a == 0) st.live[kb - k0] = (MULTI || (gq.same_x & 2u) || gq.has_imag || gq.n_h != gq.end - kb) ? (1u << P) - 1u : pairs;
    }
    __syncthreads();
    uint32_t k = k0;
    uint32_t live_next = st.live[0];  // read one group ahead: the gathers must not wait for it
    while (g < n_groups && gr.end <= k1) {  // gr = groups[g], wave-uniform
      const uint32_t live = uni(live_next);
      if (gr.x >= 256u * A && !(gr.same_x & 1u)) obs_gather<A, FAR>(cur, ps, j0, gr.x, live, far_hi);
      live_next = st.live[min(gr.end - k0, kObsChunk - 1u)];  // (past the chunk's last group: unused)
      if constexpr (MULTI) {
        const uint32_t op = uni(gr.op);
        const float upw = up ? up[op] : 0.f;  // (wave-uniform: a scalar load)
        float e = 0.f;
        obs_consume<A, true>(gr, cur, own, st, k, k0, tb, acc, upw, &e);
        if (op == 0u) ev[0] += e;  // (wave-uniform triangles: a register per observable)
        if (op == 1u) ev[1] += e;
        if (op == 2u) ev[2] += e;
        if (op == 3u) ev[3] += e;
      } else {
        if (live) obs_consume<A>(gr, cur, own, st, k, k0, tb, acc);  // else: the group vanishes on the whole block
        else k = gr.end;
      }
      ++g;
      if (g < n_groups) gr = groups[g];
    }
  }
  if (lam) {  // (null: a forward-only call that wants <psi|O|psi> alone)
    float2* ls = lam + (size_t(s_local) << n);
    typedef float v4f_nt __attribute__((ext_vector_type(4)));
    [[maybe_unused]] v4f_nt old[P];
    if constexpr (FAR) {
#pragma unroll
      for (int p = 0; p < P; ++p) old[p] = *reinterpret_cast<const v4f_nt*>(&ls[obs_phys<FAR>(j0 + 512u * p, far_hi)]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      // streamed past the L2 (QHBM_OBS_NT_STORE): lambda is not read again in this launch, and every line it would
      // occupy there is a line of psi that another workgroup is about to gather
      v4f_nt v = {acc[2 * p].x, acc[2 * p].y, acc[2 * p + 1].x, acc[2 * p + 1].y};
      if constexpr (FAR) v += old[p];
      v4f_nt* dst = reinterpret_cast<v4f_nt*>(&ls[obs_phys<FAR>(j0 + 512u * p, far_hi)]);
#if QHBM_OBS_NT_STORE
      __builtin_nontemporal_store(v, dst);
#else
      *dst = v;
#endif
    }
  }
  if constexpr (MULTI) {
    const float e0 = wave_sum(ev[0]), e1 = wave_sum(ev[1]), e2 = wave_sum(ev[2]), e3 = wave_sum(ev[3]);
    __syncthreads();  // (the staging area is free now)
    float* wave_part = reinterpret_cast<float*>(&st.term[0]);  // [4 observables][4 waves]
    if ((threadIdx.x & 63u) == 0u) {
      const uint32_t w = threadIdx.x >> 6;
      wave_part[w] = e0; wave_part[4u + w] = e1; wave_part[8u + w] = e2; wave_part[12u + w] = e3;
    }
    __syncthreads();
    // one partial per (state, logical block, observable): value_parts_multi_kernel adds them in block order
    if (threadIdx.x < n_ops && threadIdx.x < 4u) {
      const float* wp = wave_part + 4u * threadIdx.x;
      value_part[(size_t(s_local) * nb + bx) * n_ops + threadIdx.x] = (wp[0] + wp[1]) + (wp[2] + wp[3]);
    }
  }
  if constexpr (VALUE) {
    float e = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4 mine = *reinterpret_cast<const float4*>(&own[tb + 512u * p]);  // (not kept in registers: occupancy)
      e += (mine.x * acc[2 * p].x + mine.y * acc[2 * p].y) + (mine.z * acc[2 * p + 1].x + mine.w * acc[2 * p + 1].y);
    }
    e = wave_sum(e);
    // (the staging area is free now; a separate array of 16 bytes would cost a workgroup per CU)
    __syncthreads();
    float* wave_part = reinterpret_cast<float*>(&st.term[0]);
    if ((threadIdx.x & 63u) == 0) wave_part[threadIdx.x >> 6] = e;
    __syncthreads();
    // one partial per workgroup (512 atomics per state on ONE address cost a quarter of the kernel);
    // value_parts_kernel adds a state's partials in block order
    if (threadIdx.x == 0) {  // logical block: the sum order does not depend on the XCD map
      const float part = (wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]);
      float* dst = &value_part[size_t(s_local) * nb + bx];
      *dst = FAR ? *dst + part : part;  // (a far launch adds its share of <psi|O|psi> to the first launch's partial)
    }
  }
}

// <psi|O|psi> of state s = sum of its workgroups' partials, in block order (bit-reproducible), into the
// fixed-point value accumulator.
__global__ __launch_bounds__(256) void value_parts_kernel(const float* __restrict__ value_part, uint32_t n_blocks,
                                                          const float* __restrict__ op_scale,
                                                          unsigned long long* __restrict__ out64, uint32_t state0) {
  __shared__ double part[256];
  const uint32_t s = blockIdx.x;
  double acc = 0.0;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 256u) acc += double(value_part[size_t(s) * n_blocks + b]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (int(threadIdx.x) < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out64[state0 + s] += to_fixed(float(part[0]), op_scale[0]);
}

// ... the same for several observables: value_part[(state, block), op] (apply_observable_kernel<A, OBS_GATHER_MULTI>).
__global__ __launch_bounds__(256) void value_parts_multi_kernel(const float* __restrict__ value_part, uint32_t n_blocks,
                                                                uint32_t n_ops, const float* __restrict__ op_scale,
                                                                unsigned long long* __restrict__ out64, uint32_t state0) {
  __shared__ double part[256];
  const uint32_t s = blockIdx.x, t = blockIdx.y;
  double acc = 0.0;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 256u) acc += double(value_part[(size_t(s) * n_blocks + b) * n_ops + t]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (int(threadIdx.x) < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out64[size_t(state0 + s) * n_ops + t] += to_fixed(float(part[0]), op_scale[t]);
}

// ================================================================================
// Pauli terms whose X-mask flips more qubits than a tile holds (schedule.cpp global_terms): measured
// on the final state in HBM.  <psi|P|psi> = sum_j Re( i^ny (-1)^{popc(j & z)} conj(psi[j ^ x]) psi[j] ):
// j ^ x of a run of consecutive j is a permuted run, so both reads are coalesced; the partner run
// comes from wherever in the state the mask sends it (L2 / Infinity Cache).  A slow path -- one
// sweep of the state per term -- that keeps TFQ's "any PauliSum" contract (qnn.py:134-138).
// ================================================================================
__global__ __launch_bounds__(256) void measure_global_kernel(
    const float2* __restrict__ psi, uint32_t n, const DevTerm* __restrict__ terms, uint32_t n_terms,
    const float* __restrict__ op_scale, unsigned long long* __restrict__ out64, uint32_t n_ops, uint32_t state0) {
  const uint32_t s_local = blockIdx.y;
  const float2* ps = psi + (size_t(s_local) << n);
  const uint32_t j0 = blockIdx.x * 1024u + threadIdx.x;
  float2 own[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) own[a] = ps[j0 + 256u * a];
  for (uint32_t k = 0; k < n_terms; ++k) {
    const DevTerm tm = terms[k];  // wave-uniform
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint32_t j = j0 + 256u * a;
      const float2 q = ps[j ^ tm.x];
      const float wr = q.x * own[a].x + q.y * own[a].y, wi = q.x * own[a].y - q.y * own[a].x;  // conj(q) * own
      float v = (tm.ny & 1u) ? wi : wr;
      if (tm.ny == 1u || tm.ny == 2u) v = -v;
      acc += (__popc(j & tm.z) & 1) ? -v : v;
    }
    acc = wave_sum(acc * tm.coeff);
    if ((threadIdx.x & 63u) == 0)
      atomicAdd(&out64[size_t(state0 + s_local) * n_ops + tm.op], to_fixed(acc, op_scale[tm.op]));
  }
}

hipError_t launch_measure_global(const float2* psi, uint32_t n, uint32_t n_states, const DevTerm* terms,
                                 uint32_t n_terms, const float* op_scale, unsigned long long* out64, uint32_t n_ops,
                                 uint32_t state0, hipStream_t stream) {
  if (n_terms == 0 || n_states == 0) return hipSuccess;
  hipLaunchKernelGGL(measure_global_kernel, dim3((1u << n) / 1024u, n_states), dim3(256), 0, stream, psi, n, terms,
                     n_terms, op_scale, out64, n_ops, state0);
  return hipGetLastError();
}

// ================================================================================
// Per-call coefficient preparation (double precision, one thread per job).
// ================================================================================
namespace {
struct Cplx { double r, i; };
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }

// Involution G of a two-qubit kind as permutation with phases: (G psi)[i] = ph[i] * psi[perm[i]],
// matrix index = (bit_q0 << 1) | bit_q1.
__device__ void involution2(int kind, int (&perm)[4], Cplx (&ph)[4]) {
  for (int i = 0; i < 4; ++i) { perm[i] = i; ph[i] = {1.0, 0.0}; }
  switch (kind) {
    case QHBM_GATE_CNOTPOW: perm[2] = 3; perm[3] = 2; break;
    case QHBM_GATE_SWAPPOW: perm[1] = 2; perm[2] = 1; break;
    case QHBM_GATE_XXPOW: perm[0] = 3; perm[1] = 2; perm[2] = 1; perm[3] = 0; break;
    case QHBM_GATE_YYPOW:
      perm[0] = 3; perm[1] = 2; perm[2] = 1; perm[3] = 0;
      ph[0] = {-1.0, 0.0}; ph[3] = {-1.0, 0.0};
      break;
    default: break;
  }
}
}  // namespace

// blockIdx.y = program of a batch (parameter-shift): program y shifts the exponent of gate
// shift_gates[y] by shifts[y] and writes its coefficients at coef + y * coef_stride.
__global__ void prep_coefs_kernel(const CoefJob* __restrict__ jobs, int n_jobs,
                                  const float* __restrict__ params, float* __restrict__ coef,
                                  int shift_gate, double shift, const int* __restrict__ shift_gates,
                                  const float* __restrict__ shifts, uint32_t coef_stride) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  if (shift_gates) {
    shift_gate = shift_gates[blockIdx.y];
    shift = double(shifts[blockIdx.y]);
    coef += size_t(blockIdx.y) * coef_stride;
  }
  const CoefJob jb = jobs[j];
  double t = double(jb.offset);
  if (jb.param_idx >= 0) t += double(jb.scalar) * double(params[jb.param_idx]);
  if (shift_gate >= 0 && jb.gate == shift_gate) t += shift;  // (negative = unshifted, whatever the value: fixed ops carry gate -2)
  float* o = coef + jb.out_off;
  if (jb.mop == MOP_PHASE) {  // diagonal term exp(i*pi*mult*t) on its index set
    double sn, cs;
    sincospi(double(jb.mult) * t, &sn, &cs);
    o[0] = float(cs);
    o[1] = float(jb.dagger ? -sn : sn);  // adjoint plans un-apply: the conjugate phase, stored ready to use (a conjugation
    return;                              // in the kernel is an s_xor per coefficient and instance: 15 for a FULL table)
  }
  if (jb.mop == MOP_X) {  // X**t has period 2 in t: theta = pi t / 2 in [-pi/2, pi/2], |tan(theta/2)| <= 1
    t *= double(jb.mult);  // (1 except for the X**(1/2) of a lowered constant Hadamard)
    const double tr = t - 2.0 * rint(0.5 * t);
    double s2, c2;
    sincospi(0.5 * tr, &s2, &c2);
    const double sgn = jb.dagger ? -1.0 : 1.0;  // U^dagger = c*I + i*s*X: both shear coefficients negated
    o[0] = float(sgn * s2 / (1.0 + c2));  // tan(theta / 2)
    o[1] = float(sgn * s2);               // sin(theta)           (x_pair4's three shears)
    return;
  }
  double sh, ch;  // sin, cos of pi*t/2
  sincospi(0.5 * t, &sh, &ch);
  if (jb.mop == MOP_Y) {
    o[0] = float(ch);
    o[1] = float(jb.dagger ? -sh : sh);
    return;
  }
  // U = sum_k exp(i pi t e_k) P_k;   for an involution G:  U = a*I + b*G,
  // a = (1 + e^{i pi t})/2, b = (1 - e^{i pi t})/2.
  double sp, cp;
  sincospi(t, &sp, &cp);
  const Cplx A = {0.5 * (1.0 + cp), 0.5 * sp}, B = {0.5 * (1.0 - cp), -0.5 * sp};
  const double pi_d = 3.14159265358979323846;
  if (jb.mop == MOP_MAT1) {  // HPOW
    const double r = 0.70710678118654752440;
    Cplx U[4] = {{A.r + B.r * r, A.i + B.i * r}, {B.r * r, B.i * r},
                 {B.r * r, B.i * r}, {A.r - B.r * r, A.i - B.i * r}};
    for (int i = 0; i < 2; ++i)
      for (int k = 0; k < 2; ++k) {
        Cplx u = jb.dagger ? Cplx{U[k * 2 + i].r, -U[k * 2 + i].i} : U[i * 2 + k];
        o[(i * 2 + k) * 2] = float(u.r);
        o[(i * 2 + k) * 2 + 1] = float(u.i);
      }
    if (jb.dagger) {  // generator pi * H
      const double g[4] = {r, r, r, -r};
      for (int i = 0; i < 4; ++i) { o[8 + 2 * i] = float(pi_d * g[i]); o[8 + 2 * i + 1] = 0.f; }
    }
    return;
  }
  // MOP_MAT2
  Cplx U[16], Gm[16];
  for (int i = 0; i < 16; ++i) { U[i] = {0.0, 0.0}; Gm[i] = {0.0, 0.0}; }
  if (jb.op_kind == QHBM_GATE_ISWAPPOW) {
    U[0] = {1.0, 0.0}; U[15] = {1.0, 0.0};
    U[5] = {ch, 0.0}; U[10] = {ch, 0.0};
    U[6] = {0.0, sh}; U[9] = {0.0, sh};
    // dE/dt = -2 pi Im<lam|A|psi>, A = (P+ - P-)/2 = X/2 on span{01,10}
    Gm[6] = {-pi_d, 0.0}; Gm[9] = {-pi_d, 0.0};
  } else {
    int perm[4];
    Cplx ph[4];
    involution2(jb.op_kind, perm, ph);
    for (int i = 0; i < 4; ++i) {
      U[i * 4 + i].r += A.r; U[i * 4 + i].i += A.i;
      const Cplx bp = cmul(B, ph[i]);
      U[i * 4 + perm[i]].r += bp.r; U[i * 4 + perm[i]].i += bp.i;
      Gm[i * 4 + perm[i]].r += pi_d * ph[i].r;  // dE/dt = pi Im<lam|G|psi>
    }
  }
  auto idx = [&](int i) { return jb.swap ? ((i & 1) << 1) | (i >> 1) : i; };
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 4; ++k) {
      const Cplx u = jb.dagger ? Cplx{U[idx(k) * 4 + idx(i)].r, -U[idx(k) * 4 + idx(i)].i}
                               : U[idx(i) * 4 + idx(k)];
      o[(i * 4 + k) * 2] = float(u.r);
      o[(i * 4 + k) * 2 + 1] = float(u.i);
      if (jb.dagger) {
        const Cplx g = Gm[idx(i) * 4 + idx(k)];
        o[32 + (i * 4 + k) * 2] = float(g.r);
        o[32 + (i * 4 + k) * 2 + 1] = float(g.i);
      }
    }
}

// FULL[m-1] = product of the instance's PH1 / PH2 phases contained in register value m
// (program.h RecordLayout).  One block per record, run after prep_coefs_kernel.
__global__ void combine_diag_kernel(float* __restrict__ coef, const uint32_t* __restrict__ rec_offsets,
                                    int n_records, uint32_t coef_stride) {
  coef += size_t(blockIdx.y) * coef_stride;  // program of a batch
  const int r = blockIdx.x;
  const int m = threadIdx.x;
  if (r >= n_records || m == 0 || m > 15) return;
  constexpr RecordLayout L(4, false);
  float* rec = coef + rec_offsets[r];
  const uint32_t h0 = __float_as_uint(rec[0]), h1 = __float_as_uint(rec[1]);
  if (!(h1 & kFullDiagFlag)) return;
  double cr = 1.0, ci = 0.0;
  auto mul = [&](int w) {
    const double a = rec[w], b = rec[w + 1];
    const double nr = cr * a - ci * b, ni = cr * b + ci * a;
    cr = nr;
    ci = ni;
  };
  for (int j = 0; j < 4; ++j)
    if (((m >> j) & 1) && ((h0 >> (4 + j)) & 1u)) mul(L.in_ph1(j));
  for (int jb = 1; jb < 4; ++jb)
    for (int ja = 0; ja < jb; ++ja)
      if (((m >> ja) & 1) && ((m >> jb) & 1) && ((h0 >> (24 + pair_index(ja, jb))) & 1u)) mul(L.in_ph2(pair_index(ja, jb)));
  rec[L.full(m)] = float(cr);
  rec[L.full(m) + 1] = float(ci);
}

// out[s, k] = fixed-point accumulator * 2^-shift_k  (the end of every forward)
__global__ void values_from_fixed_kernel(const unsigned long long* __restrict__ acc, const float* __restrict__ inv_scale,
                                         float* __restrict__ out, uint32_t count, uint32_t n_ops) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  out[i] = float(double(static_cast<long long>(acc[i])) * double(inv_scale[i % n_ops]));
}

// grad[p] = sum_s sum_slot(p) factor * state_grad[s, slot]   (fixed order: deterministic
// given state_grad).  One block per parameter.
__global__ __launch_bounds__(256) void reduce_grad_kernel(
    const float* __restrict__ state_grad, uint32_t U, uint32_t n_slots,
    const int* __restrict__ param_slot_begin, const int* __restrict__ param_slots,
    const float* __restrict__ slot_factor, float* __restrict__ grad, int accumulate) {
  __shared__ double part[256];
  const int p = blockIdx.x;
  const int b = param_slot_begin[p], e = param_slot_begin[p + 1];
  double acc = 0.0;
  for (uint32_t s = threadIdx.x; s < U; s += 256) {
    for (int k = b; k < e; ++k) {
      const int slot = param_slots[k];
      acc += double(slot_factor[slot]) * double(state_grad[size_t(s) * n_slots + slot]);
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (int(threadIdx.x) < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) grad[p] = (accumulate ? grad[p] : 0.f) + float(part[0]);
}

// rows[s, :] *= w[s]   (the upstream weight of a single observable onto its per-state gradient slots)
__global__ void scale_rows_kernel(float* __restrict__ rows, uint32_t U, uint32_t width, const float* __restrict__ w) {
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < size_t(U) * width) rows[i] *= w[i / width];
}

// jac[s, k, p] = sum_slot(p) factor * state_grad[s, slot]   for a fixed op k
__global__ void scatter_jac_kernel(const float* __restrict__ state_grad, uint32_t U, uint32_t n_slots,
                                   const int* __restrict__ param_slot_begin,
                                   const int* __restrict__ param_slots,
                                   const float* __restrict__ slot_factor, float* __restrict__ jac,
                                   uint32_t n_ops, uint32_t op, uint32_t n_params) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= U * n_params) return;
  const uint32_t s = i / n_params, p = i % n_params;
  float acc = 0.f;
  for (int k = param_slot_begin[p]; k < param_slot_begin[p + 1]; ++k) {
    const int slot = param_slots[k];
    acc += slot_factor[slot] * state_grad[size_t(s) * n_slots + slot];
  }
  jac[(size_t(s) * n_ops + op) * n_params + p] = acc;
}

// ================================================================================
// Host-side launchers
// ================================================================================
constexpr int kMaxDevices = 64;
size_t fwd_lds_bytes(int K) { return size_t(fwd_lds(K)); }
size_t adj_lds_bytes(int K, bool exchange) { return size_t(exchange ? adjx_lds(K) : adj_lds(K)); }

// hipFuncAttributeMaxDynamicSharedMemorySize (the opt-in to more than 64 KiB of LDS) is per device:
// one flag per kernel instantiation and device of the process.
template <typename Kernel>
static hipError_t opt_in_lds(Kernel kernel, bool (&done)[kMaxDevices], size_t lds) {
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  if (dev < 0 || dev >= kMaxDevices || !done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < kMaxDevices) done[dev] = true;
  }
  return hipSuccess;
}

// Preconditions of the workgroup prologue (input_index, launched_tile, tile_base_of, relabel_lookup): its bit
// permutations are 64-lane ballots truncated to 32 bits with one table element per lane, which is correct only when
// every wave is FULL and convergent at the call (K >= 10: at least 2^(K-4) = 64 threads; the dispatch below has no
// smaller instantiation), 1 <= n_user <= 32 (`>> (32 - n_user)` and `min(L, n_user - 1)` are undefined at 0) and the
// index has at most 32 bits.  qhbm_set_circuit bounds the qubit count to [1, 31]; this guards the launch itself.
static bool pass_prologue_ok(int K, const PassArgs& a, int n_user) {
  return K >= 10 && n_user >= 1 && n_user <= 32 && a.n >= 1u && a.n <= 32u && a.n_nonlocal <= 32u;
}

template <int K, int R, bool GEN>
static hipError_t launch_fwd_t(const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits,
                               int n_user, const uint32_t* prog, const uint32_t* tables,
                               const float* coef, const float* op_scale, unsigned long long* out64, uint32_t state0,
                               hipStream_t stream, const float2* psi_src) {
  const size_t lds = fwd_lds_bytes(K);
  static bool attr_done[kMaxDevices] = {};
  if (hipError_t e = opt_in_lds(&pass_fwd_kernel<K, R, GEN>, attr_done, lds); e != hipSuccess) return e;
  const uint32_t grid = n_states << a.n_free;
  hipLaunchKernelGGL((pass_fwd_kernel<K, R, GEN>), dim3(grid), dim3(1 << (K - R)), lds, stream, a, psi, bits,
                     n_user, prog, tables, coef, op_scale, out64, state0, psi_src);
  return hipGetLastError();
}

hipError_t launch_pass_fwd(int K, int R, const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits,
                           int n_user, const uint32_t* prog, const uint32_t* tables, const float* coef,
                           const float* op_scale, unsigned long long* out64, uint32_t state0, hipStream_t stream,
                           const float2* psi_src) {
  if (!pass_prologue_ok(K, a, n_user)) return hipErrorInvalidValue;
  if (psi_src && ((a.flags & PASS_INIT_BASIS) || !a.prog_states)) return hipErrorInvalidValue;  // (batched programs that LOAD)
#define QHBM_FWD_CASE(K_, R_)                                                                          \
  if (K == K_ && R == R_)                                                                              \
    return (a.flags & PASS_GENERAL)                                                                    \
               ? launch_fwd_t<K_, R_, true>(a, n_states, psi, bits, n_user, prog, tables, coef, op_scale, out64, state0, stream, psi_src)   \
               : launch_fwd_t<K_, R_, false>(a, n_states, psi, bits, n_user, prog, tables, coef, op_scale, out64, state0, stream, psi_src);
  QHBM_FWD_CASE(10, 4)
  QHBM_FWD_CASE(11, 4)
  QHBM_FWD_CASE(12, 4)
  QHBM_FWD_CASE(13, 4)
  QHBM_FWD_CASE(14, 4)
#undef QHBM_FWD_CASE
  return hipErrorInvalidValue;
}

template <int K>
static hipError_t launch_fwd2_t(const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits, int n_user,
                                uint32_t state0, const uint32_t* prog, const uint32_t* tables, const float* coef,
                                hipStream_t stream) {
  const size_t lds = size_t(8) << K;
  static bool attr_done[kMaxDevices] = {};
  if (hipError_t e = opt_in_lds(&pass_fwd2_kernel<K>, attr_done, lds); e != hipSuccess) return e;
  const uint32_t grid = ((n_states + 1u) / 2u) << a.n_nonlocal;
  hipLaunchKernelGGL((pass_fwd2_kernel<K>), dim3(grid), dim3(1 << (K - 4)), lds, stream, a, psi, bits, n_user, state0, prog,
                     tables, coef, n_states);
  return hipGetLastError();
}

bool pass_fwd_pair_supported(int K) { return K >= 10 && K <= 13; }

hipError_t launch_pass_fwd_pair(int K, const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits, int n_user,
                                uint32_t state0, const uint32_t* prog, const uint32_t* tables, const float* coef,
                                hipStream_t stream) {
  if (!pass_prologue_ok(K, a, n_user)) return hipErrorInvalidValue;
  switch (K) {
    case 10: return launch_fwd2_t<10>(a, n_states, psi, bits, n_user, state0, prog, tables, coef, stream);
    case 11: return launch_fwd2_t<11>(a, n_states, psi, bits, n_user, state0, prog, tables, coef, stream);
    case 12: return launch_fwd2_t<12>(a, n_states, psi, bits, n_user, state0, prog, tables, coef, stream);
    case 13: return launch_fwd2_t<13>(a, n_states, psi, bits, n_user, state0, prog, tables, coef, stream);
    default: return hipErrorInvalidValue;
  }
}

// Zero fill as a KERNEL.  hipMemsetAsync becomes a memset NODE when the caller captures the call into a hipGraph, and on this
// runtime (ROCm 7.0 / gfx950) a graph warmed up on one stream and replayed on another ran its kernels without the
// memset from the second replay on (values offset by a constant: scripts/tmp/debug_capture6.py, HISTORY round 6); a
// kernel node is ordered like every other kernel of the call.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ p, size_t n16, uint32_t* __restrict__ tail,
                                                        uint32_t n_tail) {
  const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
  if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (i < n_tail) tail[i] = 0u;
}
hipError_t launch_zero_fill(void* p, size_t bytes, hipStream_t stream) {
  if (!bytes) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(p) & 15u) || (bytes & 3u)) return hipMemsetAsync(p, 0, bytes, stream);  // (never: hipMalloc'ed fp32 / u64 arrays)
  const size_t n16 = bytes / 16;
  const uint32_t n_tail = uint32_t((bytes - n16 * 16) / 4);
  const size_t threads = std::max<size_t>(n16, n_tail);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, stream, static_cast<uint4*>(p), n16,
                     reinterpret_cast<uint32_t*>(static_cast<char*>(p) + n16 * 16), n_tail);
  return hipGetLastError();
}

hipError_t launch_values_from_fixed(const unsigned long long* acc, const float* inv_scale, float* out, uint32_t count,
                                    uint32_t n_ops, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(values_from_fixed_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, acc, inv_scale, out,
                     count, n_ops);
  return hipGetLastError();
}

template <int K, bool GEN>
static hipError_t launch_adj_t(const PassArgs& a, uint32_t n_states, float2* psi, float2* lam,
                               const int8_t* bits, int n_user, const uint32_t* prog, const uint32_t* tables,
                               const float* coef, float* tile_grad, uint32_t state0, hipStream_t stream) {
  const size_t lds = adj_lds_bytes(K, false);
  static bool attr_done[kMaxDevices] = {};
  if (hipError_t e = opt_in_lds(&pass_adj_kernel<K, GEN>, attr_done, lds); e != hipSuccess) return e;
  const uint32_t grid = n_states << a.n_free;
  hipLaunchKernelGGL((pass_adj_kernel<K, GEN>), dim3(grid), dim3(1 << (K - 4)), lds, stream, a, psi, lam, bits,
                     n_user, prog, tables, coef, tile_grad, state0);
  return hipGetLastError();
}

template <int K, int ROWS>
static hipError_t launch_adjx_rows(const PassArgs& a, uint32_t n_states, float2* psi, float2* lam,
                                   const int8_t* bits, int n_user, const uint32_t* prog, const uint32_t* tables,
                                   const float* coef, float* tile_grad, uint32_t state0, hipStream_t stream) {
  const size_t lds = adj_lds_bytes(K, true);
  static bool attr_done[kMaxDevices] = {};
  if (hipError_t e = opt_in_lds(&pass_adjx_kernel<K, ROWS>, attr_done, lds); e != hipSuccess) return e;
  const uint32_t grid = n_states << a.n_free;
  hipLaunchKernelGGL((pass_adjx_kernel<K, ROWS>), dim3(grid), dim3(1 << (K - 4)), lds, stream, a, psi, lam, bits,
                     n_user, prog, tables, coef, tile_grad, state0);
#ifdef QHBM_ADJ_TIMING
  {
    unsigned long long h[8 * 64], sum[8] = {};
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_adj_phase), sizeof(h));
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 64; ++i) sum[k] += h[k * 64 + i];
    const double w = double(sum[6] ? sum[6] : 1);
    std::fprintf(stderr, "adj_timing K=%d grid=%u slots=%u waves=%llu cycles/wave: to_loads %.0f load+stage %.0f instances %.0f rounds %.0f store %.0f life %.0f\n",
                 K, grid, a.n_slots, sum[6], sum[0] / w, sum[1] / w, sum[2] / w, sum[3] / w, sum[4] / w, sum[5] / w);
    std::memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_adj_phase), h, sizeof(h));
  }
#endif
  return hipGetLastError();
}
template <int K>
static hipError_t launch_adjx_t(const PassArgs& a, uint32_t n_states, float2* psi, float2* lam,
                                const int8_t* bits, int n_user, const uint32_t* prog, const uint32_t* tables,
                                const float* coef, float* tile_grad, uint32_t state0, hipStream_t stream) {
  // (tail passes whose tiles hold no index bit 0 move their rows as 8-byte halves: the other instantiation)
  return a.c == 0 ? launch_adjx_rows<K, ROWS8>(a, n_states, psi, lam, bits, n_user, prog, tables, coef, tile_grad, state0, stream)
                  : launch_adjx_rows<K, ROWS16>(a, n_states, psi, lam, bits, n_user, prog, tables, coef, tile_grad, state0, stream);
}

hipError_t launch_pass_adj(int K, bool exchange, const PassArgs& a, uint32_t n_states, float2* psi, float2* lam,
                           const int8_t* bits, int n_user, const uint32_t* prog, const uint32_t* tables,
                           const float* coef, float* tile_grad, uint32_t state0, hipStream_t stream) {
  if (!pass_prologue_ok(K, a, n_user)) return hipErrorInvalidValue;
#define QHBM_ADJ_CASE(K_)                                                                                           \
  case K_:                                                                                                          \
    if (a.flags & PASS_GENERAL)                                                                                     \
      return launch_adj_t<K_, true>(a, n_states, psi, lam, bits, n_user, prog, tables, coef, tile_grad, state0, stream);  \
    return exchange                                                                                                 \
               ? launch_adjx_t<K_>(a, n_states, psi, lam, bits, n_user, prog, tables, coef, tile_grad, state0, stream)    \
               : launch_adj_t<K_, false>(a, n_states, psi, lam, bits, n_user, prog, tables, coef, tile_grad, state0, stream);
  switch (K) {
    QHBM_ADJ_CASE(10)
    QHBM_ADJ_CASE(11)
    QHBM_ADJ_CASE(12)
    QHBM_ADJ_CASE(13)
    default: return hipErrorInvalidValue;
  }
#undef QHBM_ADJ_CASE
}

// tile_grad holds n_states * n_tiles rows followed by scratch for the chunk sums
// (reduce_tiles_scratch_rows); the last step writes state_grad[state0 + s, slot_base + i].
size_t reduce_tiles_scratch_rows(size_t n_states, size_t n_tiles) {
  size_t rows = 0;
  for (size_t r = n_tiles; r > kTileChunk;) {
    r = (r + kTileChunk - 1) / kTileChunk;
    rows += n_states * r;
  }
  return rows;
}

hipError_t launch_reduce_tiles(float* tile_grad, uint32_t n_states, uint32_t n_tiles, uint32_t n_slots,
                               float* state_grad, uint32_t n_slots_total, uint32_t slot_base, uint32_t state0,
                               hipStream_t stream) {
  if (n_states == 0 || n_slots == 0) return hipSuccess;
  const float* src = tile_grad;
  float* scratch = tile_grad + size_t(n_states) * n_tiles * n_slots;
  uint32_t rows = n_tiles;
  const uint32_t bx = (n_slots + 63) / 64;
  while (rows > kTileChunk) {
    const uint32_t chunks = (rows + kTileChunk - 1) / kTileChunk;
    hipLaunchKernelGGL(reduce_tiles_kernel, dim3(bx, chunks, n_states), dim3(64, 16), 0, stream, src, rows, n_slots,
                       scratch, n_slots, chunks * n_slots, 0u);
    src = scratch;
    scratch += size_t(n_states) * chunks * n_slots;
    rows = chunks;
  }
  hipLaunchKernelGGL(reduce_tiles_kernel, dim3(bx, 1, n_states), dim3(64, 16), 0, stream, src, rows, n_slots,
                     state_grad + size_t(state0) * n_slots_total, 0u, n_slots_total, slot_base);
  return hipGetLastError();
}

// ================================================================================
// qhbm_statevector only: X**t and Y**t are applied as c*I - i*s*G, i.e. without cirq's global
// phase e^{i pi t / 2} (expectation values never see it), and no kernel applies a gate's
// exp(i pi t global_shift) (qhbm_gate::global_shift: -0.5 for rx / ry / rz).  The exported state
// restores the product of those phases so that it equals cirq's final_state_vector, not just its ray.
// ================================================================================
__global__ void global_phase_kernel(const CoefJob* __restrict__ jobs, int n_jobs,
                                    const ShiftPhase* __restrict__ shifts, int n_shifts,
                                    const float* __restrict__ params, float* __restrict__ out_cs) {
  __shared__ double part[256];
  double acc = 0.0;
  for (int j = threadIdx.x; j < n_jobs; j += 256) {
    const CoefJob jb = jobs[j];
    if (jb.mop != MOP_X && jb.mop != MOP_Y) continue;
    double t = double(jb.offset);
    if (jb.param_idx >= 0) t += double(jb.scalar) * double(params[jb.param_idx]);
    if (jb.mop == MOP_X) {
      t *= double(jb.mult);
      t -= 2.0 * rint(0.5 * t);  // the reduced exponent prep_coefs_kernel applies
    }
    acc += 0.5 * t;
  }
  for (int j = threadIdx.x; j < n_shifts; j += 256) {
    const ShiftPhase sp = shifts[j];
    double t = double(sp.offset);
    if (sp.param_idx >= 0) t += double(sp.scalar) * double(params[sp.param_idx]);
    acc += double(sp.shift) * t;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double sn, cs;
    sincospi(part[0], &sn, &cs);
    out_cs[0] = float(cs);
    out_cs[1] = float(sn);
  }
}

__global__ __launch_bounds__(256) void scale_states_kernel(float2* __restrict__ st, size_t count,
                                                           const float* __restrict__ cs) {
  const float c = cs[0], s = cs[1];
  for (size_t i = size_t(blockIdx.x) * 256u + threadIdx.x; i < count; i += size_t(gridDim.x) * 256u) {
    const float2 v = st[i];
    st[i] = make_float2(c * v.x - s * v.y, c * v.y + s * v.x);
  }
}

hipError_t launch_global_phase(const CoefJob* jobs, int n_jobs, const ShiftPhase* shifts, int n_shifts,
                               const float* params, float* out_cs, hipStream_t stream) {
  hipLaunchKernelGGL(global_phase_kernel, dim3(1), dim3(256), 0, stream, jobs, n_jobs, shifts, n_shifts, params,
                     out_cs);
  return hipGetLastError();
}

hipError_t launch_scale_states(float2* st, size_t count, const float* cs, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const unsigned blocks = unsigned(std::min<size_t>((count + 255) / 256, 65536));
  hipLaunchKernelGGL(scale_states_kernel, dim3(blocks), dim3(256), 0, stream, st, count, cs);
  return hipGetLastError();
}

// ================================================================================
// Sampling bitstrings from |psi|^2 (tfq.layers.Sample, qnn.py:169,177-181,286-291).
//   1. block_prob_kernel: sum of |psi|^2 over each block of 1024 amplitudes (fp64).
//   2. block_scan_kernel: per state, inclusive prefix sums of the block masses.
//   3. draw_kernel: one wave per (state, shot): Philox4x32-10 uniform -> binary search over
//      the block prefix sums -> wave prefix scan inside the block -> amplitude index -> bits.
// ================================================================================
constexpr uint32_t kSampleBlock = 1024;

__global__ __launch_bounds__(256) void block_prob_kernel(const float2* __restrict__ psi, uint32_t n,
                                                         double* __restrict__ block_mass) {
  __shared__ double part[256];
  const uint32_t nb = (1u << n) / kSampleBlock;
  const uint32_t s = blockIdx.y, b = blockIdx.x;
  const float2* p = psi + (size_t(s) << n) + size_t(b) * kSampleBlock;
  double acc = 0.0;
  for (uint32_t i = threadIdx.x; i < kSampleBlock; i += 256u) {
    const float2 v = p[i];
    acc += double(v.x) * double(v.x) + double(v.y) * double(v.y);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (int(threadIdx.x) < k) part[threadIdx.x] += part[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_mass[size_t(s) * nb + b] = part[0];
}

__global__ __launch_bounds__(256) void block_scan_kernel(double* __restrict__ block_mass, uint32_t nb) {
  __shared__ double part[256];
  __shared__ double carry;
  double* m = block_mass + size_t(blockIdx.x) * nb;
  if (threadIdx.x == 0) carry = 0.0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 256u) {
    const uint32_t i = base + threadIdx.x;
    part[threadIdx.x] = i < nb ? m[i] : 0.0;
    __syncthreads();
    for (int k = 1; k < 256; k <<= 1) {  // Hillis-Steele inclusive scan
      const double add = int(threadIdx.x) >= k ? part[threadIdx.x - k] : 0.0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) m[i] = carry + part[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 255) carry += part[255];
    __syncthreads();
  }
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = uint64_t(0xD2511F53u) * c[0], p1 = uint64_t(0xCD9E8D57u) * c[2];
    const uint32_t n0 = uint32_t(p1 >> 32) ^ c[1] ^ k0, n2 = uint32_t(p0 >> 32) ^ c[3] ^ k1;
    c[1] = uint32_t(p1);
    c[3] = uint32_t(p0);
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(64) void draw_kernel(const float2* __restrict__ psi, uint32_t n, int n_user,
                                                  const double* __restrict__ block_cum, uint32_t n_shots,
                                                  uint64_t seed, uint32_t state0, int8_t* __restrict__ out) {
  const uint32_t shot = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
  const uint32_t nb = (1u << n) / kSampleBlock;
  const double* cum = block_cum + size_t(s) * nb;
  uint32_t c[4] = {shot, state0 + s, 0x51b0c6a1u, 0u};
  philox4x32_10(c, uint32_t(seed), uint32_t(seed >> 32));
  const double total = cum[nb - 1];
  const double u = (double(c[0]) * 0x1p-32 + double(c[1]) * 0x1p-64) * total;  // [0, total)
  uint32_t lo = 0, hi = nb - 1;  // first block whose inclusive prefix exceeds u
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (cum[mid] > u) hi = mid; else lo = mid + 1;
  }
  const float r = float(u - (lo ? cum[lo - 1] : 0.0));
  const float2* p = psi + (size_t(s) << n) + size_t(lo) * kSampleBlock;
  constexpr int kPer = kSampleBlock / 64;  // 16 consecutive amplitudes per lane
  float w[kPer];
  float mine = 0.f;
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const float2 v = p[lane * kPer + i];
    w[i] = v.x * v.x + v.y * v.y;
    mine += w[i];
  }
  float incl = mine;  // inclusive prefix over lanes
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const float t = __shfl_up(incl, k);
    if (int(lane) >= k) incl += t;
  }
  const float excl = incl - mine;
  // the lane whose interval [excl, incl) holds r; rounding may push r past the block mass:
  // then the last lane with any mass takes it
  const bool hit = (r >= excl && r < incl) || (lane == 63 && r >= incl);
  uint64_t ballot = __ballot(hit && mine > 0.f);
  if (ballot == 0) ballot = __ballot(mine > 0.f);
  if (ballot == 0) ballot = 1ull;
  const int owner = (r >= __shfl(incl, 63)) ? 63 - __builtin_clzll(ballot) : __builtin_ctzll(ballot);
  if (int(lane) == owner) {
    float acc = excl;
    int pick = -1, last_nz = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      if (w[i] > 0.f) last_nz = i;
      acc += w[i];
      if (pick < 0 && r < acc && w[i] > 0.f) pick = i;
    }
    if (pick < 0) pick = last_nz;
    const uint32_t idx = lo * kSampleBlock + lane * kPer + uint32_t(pick);
    int8_t* o = out + (size_t(state0 + s) * n_shots + shot) * size_t(n_user);
    for (int q = 0; q < n_user; ++q) o[q] = int8_t((idx >> (n_user - 1 - q)) & 1u);
  }
}

hipError_t launch_sample(const float2* psi, uint32_t n, int n_user, uint32_t n_states, double* block_cum,
                         uint32_t n_shots, uint64_t seed, uint32_t state0, int8_t* out, hipStream_t stream) {
  const uint32_t nb = (1u << n) / kSampleBlock;
  hipLaunchKernelGGL(block_prob_kernel, dim3(nb, n_states), dim3(256), 0, stream, psi, n, block_cum);
  hipLaunchKernelGGL(block_scan_kernel, dim3(n_states), dim3(256), 0, stream, block_cum, nb);
  if (n_shots) hipLaunchKernelGGL(draw_kernel, dim3(n_shots, n_states), dim3(64), 0, stream, psi, n, n_user, block_cum,
                                  n_shots, seed, state0, out);
  return hipGetLastError();
}

// ================================================================================
// Shot COUNTS per outcome for a batch of (program, state) elements (qhbm_sample_counts): what the
// sampled estimators of qnn.py:170-226 reduce their shots to -- a Pauli-string estimate is a signed sum
// of the counts, an energy average a weighted one -- without materialising n_shots bitstrings per
// element.  Element e = program q = e / prog_states on state state0 + e % prog_states; shot j is drawn
// with Philox keyed by (seed; j, state, program), so counts are reproducible and independent of how
// the batch is cut into launch sets.  Integer atomics only: bit-reproducible.
//   n <= 10 (one 1024-amplitude block per element): one workgroup per element builds the cumulative
//   distribution in LDS (fp64) and every THREAD draws shots by binary search -- 64 x the shot rate of the
//   wave-per-shot kernel, which is what a million shots per element need;
//   n  > 10: the block-mass / prefix / wave-per-shot pipeline of qhbm_sample with a global counter per outcome.
// ================================================================================
__global__ __launch_bounds__(256) void sample_counts_small_kernel(const float2* __restrict__ psi, int n_user,
                                                                  uint32_t prog_states, uint32_t prog0, uint32_t n_shots,
                                                                  uint64_t seed, uint32_t state0, uint32_t n_states_total,
                                                                  int* __restrict__ out /*[programs, n_states_total, 2^n_user]*/) {
  __shared__ double cum[kSampleBlock];
  __shared__ double part[256];
  __shared__ int hist[kSampleBlock];
  const uint32_t e = blockIdx.x, tid = threadIdx.x;
  const uint32_t q = prog0 + e / prog_states, srow = state0 + e % prog_states;
  const float2* p = psi + size_t(e) * kSampleBlock;
  double w[4], mine = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 v = p[tid * 4u + uint32_t(i)];
    w[i] = double(v.x) * double(v.x) + double(v.y) * double(v.y);
    mine += w[i];
    hist[tid * 4u + uint32_t(i)] = 0;
  }
  part[tid] = mine;
  __syncthreads();
  for (int k = 1; k < 256; k <<= 1) {  // Hillis-Steele inclusive scan of the thread sums
    const double add = int(tid) >= k ? part[tid - k] : 0.0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  double run = part[tid] - mine;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    run += w[i];
    cum[tid * 4u + uint32_t(i)] = run;
  }
  __syncthreads();
  const double total = cum[kSampleBlock - 1];
  for (uint32_t shot = tid; shot < n_shots; shot += 256u) {
    uint32_t c[4] = {shot, srow, 0x51b0c6a1u, q};
    philox4x32_10(c, uint32_t(seed), uint32_t(seed >> 32));
    const double u = (double(c[0]) * 0x1p-32 + double(c[1]) * 0x1p-64) * total;  // [0, total)
    uint32_t lo = 0, hi = kSampleBlock - 1;  // first outcome whose inclusive prefix exceeds u (it has mass)
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cum[mid] > u) hi = mid; else lo = mid + 1;
    }
    atomicAdd(&hist[lo], 1);
  }
  __syncthreads();
  const uint32_t dim = 1u << n_user;  // idle padding qubits are the high index bits and stay |0>
  int* o = out + (size_t(q) * n_states_total + srow) * dim;
  for (uint32_t x = tid; x < dim; x += 256u) o[x] = hist[x];
}

__global__ __launch_bounds__(64) void draw_counts_kernel(const float2* __restrict__ psi, uint32_t n, int n_user,
                                                         const double* __restrict__ block_cum, uint32_t prog_states,
                                                         uint32_t prog0, uint64_t seed, uint32_t state0,
                                                         uint32_t n_states_total, int* __restrict__ out) {
  const uint32_t shot = blockIdx.x, e = blockIdx.y, lane = threadIdx.x;
  const uint32_t q = prog0 + e / prog_states, srow = state0 + e % prog_states;
  const uint32_t nb = (1u << n) / kSampleBlock;
  const double* cum = block_cum + size_t(e) * nb;
  uint32_t c[4] = {shot, srow, 0x51b0c6a1u, q};
  philox4x32_10(c, uint32_t(seed), uint32_t(seed >> 32));
  const double total = cum[nb - 1];
  const double u = (double(c[0]) * 0x1p-32 + double(c[1]) * 0x1p-64) * total;
  uint32_t lo = 0, hi = nb - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (cum[mid] > u) hi = mid; else lo = mid + 1;
  }
  const float r = float(u - (lo ? cum[lo - 1] : 0.0));
  const float2* p = psi + (size_t(e) << n) + size_t(lo) * kSampleBlock;
  constexpr int kPer = kSampleBlock / 64;
  float w[kPer];
  float mine = 0.f;
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const float2 v = p[lane * kPer + i];
    w[i] = v.x * v.x + v.y * v.y;
    mine += w[i];
  }
  float incl = mine;
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const float t = __shfl_up(incl, k);
    if (int(lane) >= k) incl += t;
  }
  const float excl = incl - mine;
  const bool hit = (r >= excl && r < incl) || (lane == 63 && r >= incl);
  uint64_t ballot = __ballot(hit && mine > 0.f);
  if (ballot == 0) ballot = __ballot(mine > 0.f);
  if (ballot == 0) ballot = 1ull;
  const int owner = (r >= __shfl(incl, 63)) ? 63 - __builtin_clzll(ballot) : __builtin_ctzll(ballot);
  if (int(lane) == owner) {
    float acc = excl;
    int pick = -1, last_nz = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      if (w[i] > 0.f) last_nz = i;
      acc += w[i];
      if (pick < 0 && r < acc && w[i] > 0.f) pick = i;
    }
    if (pick < 0) pick = last_nz;
    const uint32_t idx = lo * kSampleBlock + lane * kPer + uint32_t(pick);
    atomicAdd(&out[((size_t(q) * n_states_total + srow) << n_user) + (idx & ((1u << n_user) - 1u))], 1);
  }
}

hipError_t launch_sample_counts(const float2* psi, uint32_t n, int n_user, uint32_t n_elements, uint32_t prog_states,
                                uint32_t prog0, double* block_cum, uint32_t n_shots, uint64_t seed, uint32_t state0,
                                uint32_t n_states_total, int* out, hipStream_t stream) {
  if (!n_elements) return hipSuccess;
  if (n == uint32_t(kMinTileBits)) {
    hipLaunchKernelGGL(sample_counts_small_kernel, dim3(n_elements), dim3(256), 0, stream, psi, n_user, prog_states, prog0,
                       n_shots, seed, state0, n_states_total, out);
    return hipGetLastError();
  }
  if (n_elements > 65535u) return hipErrorInvalidValue;  // grid.y; the engine cuts its launch sets accordingly
  const uint32_t nb = (1u << n) / kSampleBlock;  // (the caller zeroed `out`: this path adds to global counters)
  hipLaunchKernelGGL(block_prob_kernel, dim3(nb, n_elements), dim3(256), 0, stream, psi, n, block_cum);
  hipLaunchKernelGGL(block_scan_kernel, dim3(n_elements), dim3(256), 0, stream, block_cum, nb);
  if (n_shots)
    hipLaunchKernelGGL(draw_counts_kernel, dim3(n_shots, n_elements), dim3(64), 0, stream, psi, n, n_user, block_cum,
                       prog_states, prog0, seed, state0, n_states_total, out);
  return hipGetLastError();
}

// ================================================================================
// EBM side: spin-parity energies of bitstrings,
//   E(x) = sum_k theta_k * prod_{q in S_k} (1 - 2 x_q)
// BernoulliEnergy / KOBE = SpinsFromBitstrings -> Parity -> VariableDot
// (qhbmlib/models/energy.py:123-209, energy_utils.py:39-110).  S_k is a bit mask over the
// columns of the bitstring.  One thread per bitstring; terms are staged in LDS.
// ================================================================================
constexpr int kParityChunk = 1024;  // terms staged per LDS fill

__device__ __forceinline__ uint64_t pack_bits(const int8_t* __restrict__ row, int n) {
  uint64_t x = 0;
  for (int q = 0; q < n; ++q) x |= uint64_t(row[q] & 1) << q;
  return x;
}

__global__ __launch_bounds__(256) void parity_energy_kernel(const int8_t* __restrict__ bits, int64_t n_rows, int n,
                                                            const uint64_t* __restrict__ masks,
                                                            const float* __restrict__ thetas, int n_terms,
                                                            float* __restrict__ energy) {
  __shared__ uint64_t sm[kParityChunk];
  __shared__ float st[kParityChunk];
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const uint64_t x = i < n_rows ? pack_bits(bits + i * n, n) : 0ull;
  float e = 0.f;
  for (int k0 = 0; k0 < n_terms; k0 += kParityChunk) {
    const int kn = min(kParityChunk, n_terms - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < kn; k += 256) { sm[k] = masks[k0 + k]; st[k] = thetas[k0 + k]; }
    __syncthreads();
    for (int k = 0; k < kn; ++k) e += (__popcll(x & sm[k]) & 1) ? -st[k] : st[k];
  }
  if (i < n_rows) energy[i] = e;
}

// grad[k] = sum_i w[i] * parity_k(x_i): the VJP of the energies with respect to theta.
// A workgroup keeps 256 x 8 bitstrings (and their weights) in registers and walks the terms.
constexpr int kParityRows = 8;
__global__ __launch_bounds__(256) void parity_energy_vjp_kernel(const int8_t* __restrict__ bits, int64_t n_rows, int n,
                                                                const uint64_t* __restrict__ masks, int n_terms,
                                                                const float* __restrict__ w,
                                                                float* __restrict__ grad) {
  uint64_t x[kParityRows];
  float wt[kParityRows];
#pragma unroll
  for (int j = 0; j < kParityRows; ++j) {
    const int64_t i = (int64_t(blockIdx.x) * kParityRows + j) * 256 + threadIdx.x;
    x[j] = i < n_rows ? pack_bits(bits + i * n, n) : 0ull;
    wt[j] = i < n_rows ? w[i] : 0.f;
  }
  __shared__ float part[4];
  for (int k = 0; k < n_terms; ++k) {
    const uint64_t m = masks[k];
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < kParityRows; ++j) acc += (__popcll(x[j] & m) & 1) ? -wt[j] : wt[j];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float v = part[0] + part[1] + part[2] + part[3];
      if (v != 0.f) atomicAdd(&grad[k], v);
    }
    __syncthreads();
  }
}

hipError_t launch_parity_energy(const int8_t* bits, int64_t n_rows, int n, const uint64_t* masks,
                                const float* thetas, int n_terms, float* energy, hipStream_t stream) {
  if (n_rows == 0) return hipSuccess;
  hipLaunchKernelGGL(parity_energy_kernel, dim3(unsigned((n_rows + 255) / 256)), dim3(256), 0, stream, bits, n_rows, n,
                     masks, thetas, n_terms, energy);
  return hipGetLastError();
}

hipError_t launch_parity_energy_vjp(const int8_t* bits, int64_t n_rows, int n, const uint64_t* masks, int n_terms,
                                    const float* w, float* grad, hipStream_t stream) {
  if (n_rows == 0 || n_terms == 0) return hipSuccess;
  const int64_t per_block = 256 * kParityRows;
  hipLaunchKernelGGL(parity_energy_vjp_kernel, dim3(unsigned((n_rows + per_block - 1) / per_block)), dim3(256), 0,
                     stream, bits, n_rows, n, masks, n_terms, w, grad);
  return hipGetLastError();
}

size_t observable_value_parts(uint32_t n, uint32_t n_states) { return size_t(n_states) * ((1u << n) / (256u * obs_amps_per_thread(n))); }

hipError_t launch_apply_observable(const float2* psi, float2* lam, uint32_t n, uint32_t n_states,
                                   const DevTerm* terms, uint32_t n_terms, const ObsGroup* groups,
                                   uint32_t n_groups, const float* upstream, uint32_t n_ops, uint32_t state0,
                                   const float* op_scale, unsigned long long* out64, float* value_part,
                                   bool xcd_states, hipStream_t stream, bool multi, const ObsFarLaunch* far, int n_far) {
  // out64: single observable -- unweighted lambda + <psi|O|psi>; `multi` -- the weighted lambda and every value
  const int mode = multi ? OBS_GATHER_MULTI : (out64 != nullptr ? OBS_GATHER_VALUE : OBS_GATHER_LAMBDA);
  if (multi && (n_ops < 2u || n_ops > kObsGatherMultiOps || !out64)) return hipErrorInvalidValue;
  const uint32_t nb = (1u << n) / (256u * obs_amps_per_thread(n));
  const uint32_t xs = xcd_states && nb >= 128u ? 1u : 0u;  // (a state must at least fill an XCD's workgroup slots)
#define QHBM_OBS(A_, V_)                                                                                          \
  hipLaunchKernelGGL((apply_observable_kernel<A_, V_>), dim3(nb * n_states), dim3(256), 0, stream, psi, lam, n, terms, \
                     n_terms, groups, n_groups, upstream, n_ops, state0, value_part, nb, n_states, xs, 0u)
  if (obs_amps_per_thread(n) == 8u) {
    if (mode == OBS_GATHER_MULTI) QHBM_OBS(8, OBS_GATHER_MULTI);
    else if (mode == OBS_GATHER_VALUE) QHBM_OBS(8, OBS_GATHER_VALUE);
    else QHBM_OBS(8, OBS_GATHER_LAMBDA);
  } else {
    if (mode == OBS_GATHER_MULTI) QHBM_OBS(4, OBS_GATHER_MULTI);
    else if (mode == OBS_GATHER_VALUE) QHBM_OBS(4, OBS_GATHER_VALUE);
    else QHBM_OBS(4, OBS_GATHER_LAMBDA);
  }
#undef QHBM_OBS
  // the far launches of the two-level sweep (single-observable modes, eight amplitudes per thread): lambda and the value
  // partials accumulate onto the first launch's
  for (int f = 0; f < n_far; ++f) {
    if (multi || obs_amps_per_thread(n) != 8u || far[f].far_hi < 11u || far[f].far_hi + 7u > n) return hipErrorInvalidValue;
    if (mode == OBS_GATHER_VALUE)
      hipLaunchKernelGGL((apply_observable_kernel<8, OBS_GATHER_VALUE, true>), dim3(nb * n_states), dim3(256), 0, stream, psi, lam, n,
                         far[f].terms, far[f].n_terms, far[f].groups, far[f].n_groups, upstream, n_ops, state0, value_part, nb,
                         n_states, xs, far[f].far_hi);
    else
      hipLaunchKernelGGL((apply_observable_kernel<8, OBS_GATHER_LAMBDA, true>), dim3(nb * n_states), dim3(256), 0, stream, psi, lam, n,
                         far[f].terms, far[f].n_terms, far[f].groups, far[f].n_groups, upstream, n_ops, state0, value_part, nb,
                         n_states, xs, far[f].far_hi);
  }
  if (mode == OBS_GATHER_VALUE && n_states)
    hipLaunchKernelGGL(value_parts_kernel, dim3(n_states), dim3(256), 0, stream, value_part, nb, op_scale, out64, state0);
  if (mode == OBS_GATHER_MULTI && n_states)
    hipLaunchKernelGGL(value_parts_multi_kernel, dim3(n_states, n_ops), dim3(256), 0, stream, value_part, nb, n_ops, op_scale,
                       out64, state0);
  return hipGetLastError();
}

hipError_t launch_prep_coefs(const CoefJob* jobs, int n_jobs, const float* params, float* coef,
                             int shift_gate, double shift, hipStream_t stream) {
  if (n_jobs == 0) return hipSuccess;
  hipLaunchKernelGGL(prep_coefs_kernel, dim3((n_jobs + 127) / 128), dim3(128), 0, stream, jobs, n_jobs,
                     params, coef, shift_gate, shift, static_cast<const int*>(nullptr),
                     static_cast<const float*>(nullptr), 0u);
  return hipGetLastError();
}

hipError_t launch_prep_coefs_batch(const CoefJob* jobs, int n_jobs, const float* params, float* coef,
                                   const int* shift_gates, const float* shifts, uint32_t n_programs,
                                   uint32_t coef_stride, hipStream_t stream) {
  if (n_jobs == 0 || n_programs == 0) return hipSuccess;
  hipLaunchKernelGGL(prep_coefs_kernel, dim3((n_jobs + 127) / 128, n_programs), dim3(128), 0, stream, jobs, n_jobs,
                     params, coef, -1, 0.0, shift_gates, shifts, coef_stride);
  return hipGetLastError();
}

hipError_t launch_combine_diag(float* coef, const uint32_t* rec_offsets, int n_records, uint32_t n_programs,
                               uint32_t coef_stride, hipStream_t stream) {
  if (n_records == 0 || n_programs == 0) return hipSuccess;
  hipLaunchKernelGGL(combine_diag_kernel, dim3(n_records, n_programs), dim3(16), 0, stream, coef, rec_offsets,
                     n_records, coef_stride);
  return hipGetLastError();
}

// dst[y * stride + i] = src[i]: the static words of a plan's coefficient buffer, once per program copy
__global__ void replicate_kernel(const float* __restrict__ src, float* __restrict__ dst, uint32_t words, uint32_t stride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) dst[size_t(blockIdx.y) * stride + i] = src[i];
}
hipError_t launch_replicate(const float* src, float* dst, uint32_t words, uint32_t stride, uint32_t copies,
                            hipStream_t stream) {
  if (!words || !copies) return hipSuccess;
  hipLaunchKernelGGL(replicate_kernel, dim3((words + 255) / 256, copies), dim3(256), 0, stream, src, dst, words, stride);
  return hipGetLastError();
}

// Parameter-shift bookkeeping.  prog_acc[q0 + q] += sum_{u, k} upstream[s0 + u, k] * vals[q, u, k]
// (one block per program of the batch; double accumulation in a fixed order) ...
__global__ __launch_bounds__(256) void shift_program_accumulate_kernel(const float* __restrict__ vals,
                                                                       const float* __restrict__ upstream, uint32_t c,
                                                                       uint32_t n_ops, uint32_t s0,
                                                                       double* __restrict__ prog_acc,
                                                                       const int* __restrict__ dst_index) {
  __shared__ double part[256];
  const float* v = vals + size_t(blockIdx.x) * c * n_ops;
  const float* up = upstream + size_t(s0) * n_ops;
  double acc = 0.0;
  for (uint32_t i = threadIdx.x; i < c * n_ops; i += 256) acc += double(up[i]) * double(v[i]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (int(threadIdx.x) < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  // (dst_index: the programs run sorted by the pass their shifted gate sits in; the sums land in gate order)
  if (threadIdx.x == 0) prog_acc[dst_index ? uint32_t(dst_index[blockIdx.x]) : blockIdx.x] += part[0];
}
// ... and grad[p] = sum over the gates g driven by p (in gate order) of weight_g * (acc[2g] - acc[2g+1]).
__global__ void shift_combine_kernel(const double* __restrict__ prog_acc, const int* __restrict__ gate_param,
                                     const float* __restrict__ gate_weight, int n_shift_gates, float* __restrict__ grad,
                                     int n_params) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_params) return;
  double acc = 0.0;
  for (int g = 0; g < n_shift_gates; ++g)
    if (gate_param[g] == p) acc += double(gate_weight[g]) * (prog_acc[2 * g] - prog_acc[2 * g + 1]);
  grad[p] = float(acc);
}
hipError_t launch_shift_program_accumulate(const float* vals, const float* upstream, uint32_t n_programs, uint32_t c,
                                           uint32_t n_ops, uint32_t s0, double* prog_acc, hipStream_t stream,
                                           const int* dst_index) {
  if (!n_programs) return hipSuccess;
  hipLaunchKernelGGL(shift_program_accumulate_kernel, dim3(n_programs), dim3(256), 0, stream, vals, upstream, c, n_ops,
                     s0, prog_acc, dst_index);
  return hipGetLastError();
}
hipError_t launch_shift_combine(const double* prog_acc, const int* gate_param, const float* gate_weight,
                                int n_shift_gates, float* grad, int n_params, hipStream_t stream) {
  if (!n_params) return hipSuccess;
  hipLaunchKernelGGL(shift_combine_kernel, dim3((n_params + 127) / 128), dim3(128), 0, stream, prog_acc, gate_param,
                     gate_weight, n_shift_gates, grad, n_params);
  return hipGetLastError();
}

hipError_t launch_reduce_grad(const float* state_grad, uint32_t U, uint32_t n_slots,
                              const int* param_slot_begin, const int* param_slots,
                              const float* slot_factor, float* grad, int n_params, int accumulate,
                              hipStream_t stream) {
  if (n_params == 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_grad_kernel, dim3(n_params), dim3(256), 0, stream, state_grad, U, n_slots,
                     param_slot_begin, param_slots, slot_factor, grad, accumulate);
  return hipGetLastError();
}

// ---- sustained packed-fp32 rate and shader clock (qhbm_clock_probe; bench.py roofline.compute.attainable_peak) ----
// Every wave runs kProbeIters x 16 independent v_pk_fma_f32 with a scalar coefficient operand (the form of the pass
// kernels) between two s_memtime (shader cycles) / s_memrealtime (100 MHz) reads; one workgroup of 1024 threads per
// CU: four waves per SIMD.  out[wave] = (cycles, real-time ticks).
constexpr int kProbeIters = 4096;
__global__ __launch_bounds__(1024) void clock_probe_kernel(uint64_t* __restrict__ out, float x, float* __restrict__ sink) {
  v2f a0{x, 1.f}, a1{x, 2.f}, a2{x, 3.f}, a3{x, 4.f}, a4{x, 5.f}, a5{x, 6.f}, a6{x, 7.f}, a7{x, 8.f};
  const v2f c{1.0001f, 0.0001f};
  uint64_t t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0));
  for (int i = 0; i < kProbeIters; ++i) {
#define QHBM_PB(A_) "v_pk_fma_f32 %" #A_ ", %" #A_ ", %8, %" #A_ "\n\t"
    asm volatile(QHBM_PB(0) QHBM_PB(1) QHBM_PB(2) QHBM_PB(3) QHBM_PB(4) QHBM_PB(5) QHBM_PB(6) QHBM_PB(7)
                 QHBM_PB(0) QHBM_PB(1) QHBM_PB(2) QHBM_PB(3) QHBM_PB(4) QHBM_PB(5) QHBM_PB(6) "v_pk_fma_f32 %7, %7, %8, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(c));
#undef QHBM_PB
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63u) == 0u) { out[2 * gw] = t1 - t0; out[2 * gw + 1] = r1 - r0; }
  const v2f sum = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (sum.x == 12345.678f) sink[0] = sum.y;  // (keeps the arithmetic alive)
}
hipError_t launch_clock_probe(uint64_t* out, float* sink, uint32_t n_cus, hipStream_t stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(n_cus), dim3(1024), 0, stream, out, 1.0f, sink);
  return hipGetLastError();
}
uint32_t clock_probe_waves(uint32_t n_cus) { return n_cus * 16u; }
double clock_probe_instructions_per_simd() { return double(kProbeIters) * 16.0 * 4.0; }

hipError_t launch_scale_rows(float* rows, uint32_t U, uint32_t width, const float* w, hipStream_t stream) {
  const size_t total = size_t(U) * width;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, stream, rows, U, width, w);
  return hipGetLastError();
}

hipError_t launch_scatter_jac(const float* state_grad, uint32_t U, uint32_t n_slots,
                              const int* param_slot_begin, const int* param_slots,
                              const float* slot_factor, float* jac, uint32_t n_ops, uint32_t op,
                              uint32_t n_params, hipStream_t stream) {
  const uint32_t total = U * n_params;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_jac_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, state_grad, U,
                     n_slots, param_slot_begin, param_slots, slot_factor, jac, n_ops, op, n_params);
  return hipGetLastError();
}

}  // namespace qhbm
