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

#include "../../include/qhbm_engine.h"
#include "kernels.h"
#include "program.h"

namespace qhbm {

namespace {

constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ uint32_t swz(uint32_t l) { return l ^ ((l >> 5) & 31u); }

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int N> struct IC { static constexpr int value = N; };

// Waves per SIMD the register allocator must leave room for: as many workgroups per CU
// as the LDS footprint admits (160 KiB per CU), capped at 4 waves per SIMD.
constexpr int wg_per_cu(int lds_bytes) { return (160 * 1024) / lds_bytes < 1 ? 1 : (160 * 1024) / lds_bytes; }
constexpr int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
constexpr int fwd_min_waves(int K, int R) {
  return clampi(wg_per_cu((8 << K) + 8192) * (1 << (K - R)) / 256, 1, 4);
}
constexpr int adj_min_waves(int K) {
  return clampi(wg_per_cu((16 << K) + 12288) * (1 << (K - 4)) / 256, 1, 4);
}

// Static dispatch on a wave-uniform register-bit index (scalar branch).
#define QHBM_DISPATCH_RB(R_, rb_, CALL_)                \
  switch (rb_) {                                        \
    case 0: { constexpr int RB = 0; CALL_; } break;     \
    case 1: { constexpr int RB = 1; CALL_; } break;     \
    case 2: { constexpr int RB = 2; CALL_; } break;     \
    case 3: { constexpr int RB = 3; CALL_; } break;     \
    default:                                            \
      if constexpr ((R_) > 4) { constexpr int RB = (R_) > 4 ? 4 : 0; CALL_; } \
      break;                                            \
  }

// ---- in-register gate kernels --------------------------------------------------
// c*I - i*s*X  on register bit RB
template <int R, int RB>
__device__ __forceinline__ void apply_x(float (&ar)[1 << R], float (&ai)[1 << R], float c, float s) {
#pragma unroll
  for (int p = 0; p < (1 << (R - 1)); ++p) {
    const int m = ((p >> RB) << (RB + 1)) | (p & ((1 << RB) - 1));
    const int m1 = m | (1 << RB);
    const float r0 = ar[m], i0 = ai[m], r1 = ar[m1], i1 = ai[m1];
    ar[m] = fmaf(s, i1, c * r0);
    ai[m] = fmaf(-s, r1, c * i0);
    ar[m1] = fmaf(s, i0, c * r1);
    ai[m1] = fmaf(-s, r0, c * i1);
  }
}

// c*I - i*s*Y = [[c, -s], [s, c]]
template <int R, int RB>
__device__ __forceinline__ void apply_y(float (&ar)[1 << R], float (&ai)[1 << R], float c, float s) {
#pragma unroll
  for (int p = 0; p < (1 << (R - 1)); ++p) {
    const int m = ((p >> RB) << (RB + 1)) | (p & ((1 << RB) - 1));
    const int m1 = m | (1 << RB);
    const float r0 = ar[m], i0 = ai[m], r1 = ar[m1], i1 = ai[m1];
    ar[m] = fmaf(-s, r1, c * r0);
    ai[m] = fmaf(-s, i1, c * i0);
    ar[m1] = fmaf(s, r0, c * r1);
    ai[m1] = fmaf(s, i0, c * i1);
  }
}

// general 2x2, u = row-major {re, im} x 4 (wave-uniform)
template <int R, int RB>
__device__ __forceinline__ void apply_mat1(float (&ar)[1 << R], float (&ai)[1 << R], const float* u) {
#pragma unroll
  for (int p = 0; p < (1 << (R - 1)); ++p) {
    const int m = ((p >> RB) << (RB + 1)) | (p & ((1 << RB) - 1));
    const int m1 = m | (1 << RB);
    const float r0 = ar[m], i0 = ai[m], r1 = ar[m1], i1 = ai[m1];
    ar[m] = u[0] * r0 - u[1] * i0 + u[2] * r1 - u[3] * i1;
    ai[m] = u[0] * i0 + u[1] * r0 + u[2] * i1 + u[3] * r1;
    ar[m1] = u[4] * r0 - u[5] * i0 + u[6] * r1 - u[7] * i1;
    ai[m1] = u[4] * i0 + u[5] * r0 + u[6] * i1 + u[7] * r1;
  }
}

// Im <lam| G |psi> restricted to this thread's registers, for the generators
// of the fast-path gates.  X: pairs swap.  Y: (Y psi)_0 = -i psi_1, (Y psi)_1 = i psi_0.
template <int R, int RB>
__device__ __forceinline__ float im_lam_x_psi(const float (&pr)[1 << R], const float (&pi)[1 << R],
                                              const float (&lr)[1 << R], const float (&li)[1 << R]) {
  float acc = 0.f;
#pragma unroll
  for (int m = 0; m < (1 << R); ++m) {
    const int m1 = m ^ (1 << RB);
    // Im(conj(lam_m) * psi_m1) = lr*pi - li*pr
    acc += lr[m] * pi[m1] - li[m] * pr[m1];
  }
  return acc;
}
template <int R, int RB>
__device__ __forceinline__ float im_lam_y_psi(const float (&pr)[1 << R], const float (&pi)[1 << R],
                                              const float (&lr)[1 << R], const float (&li)[1 << R]) {
  float acc = 0.f;
#pragma unroll
  for (int p = 0; p < (1 << (R - 1)); ++p) {
    const int m = ((p >> RB) << (RB + 1)) | (p & ((1 << RB) - 1));
    const int m1 = m | (1 << RB);
    // conj(l0)*(-i p1) + conj(l1)*(i p0); Im(conj(l)*(-i p)) = -Re(conj(l) p), Im(conj(l)*(i p)) = Re(conj(l) p)
    acc += -(lr[m] * pr[m1] + li[m] * pi[m1]) + (lr[m1] * pr[m] + li[m1] * pi[m]);
  }
  return acc;
}
// Im sum_ij conj(lam_i) g_ij psi_j over pairs / quads, g wave-uniform row-major complex
template <int R, int RB>
__device__ __forceinline__ float im_lam_g1_psi(const float (&pr)[1 << R], const float (&pi)[1 << R],
                                               const float (&lr)[1 << R], const float (&li)[1 << R],
                                               const float* g) {
  float acc = 0.f;
#pragma unroll
  for (int p = 0; p < (1 << (R - 1)); ++p) {
    const int m = ((p >> RB) << (RB + 1)) | (p & ((1 << RB) - 1));
    const int ix[2] = {m, m | (1 << RB)};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float sr = 0.f, si = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float gr = g[(i * 2 + j) * 2], gi = g[(i * 2 + j) * 2 + 1];
        sr += gr * pr[ix[j]] - gi * pi[ix[j]];
        si += gr * pi[ix[j]] + gi * pr[ix[j]];
      }
      acc += lr[ix[i]] * si - li[ix[i]] * sr;
    }
  }
  return acc;
}
// ---- shared pieces of the pass kernels ----------------------------------------
struct TileCtx {
  uint32_t tile_base;  // nonlocal bits of this tile, in index space
  uint32_t cmask;
  uint32_t c;
  const uint32_t* spread;
};

__device__ __forceinline__ uint32_t global_index(const TileCtx& t, uint32_t l) {
  return t.tile_base | (l & t.cmask) | t.spread[l >> t.c];
}

template <int K, int NT>
__device__ __forceinline__ void load_tile(float2* __restrict__ tile, const float2* __restrict__ st,
                                          const TileCtx& t, int tid) {
#pragma unroll 4
  for (int p = tid; p < (1 << (K - 1)); p += NT) {
    const uint32_t l = 2u * p;
    const float4 v = *reinterpret_cast<const float4*>(st + global_index(t, l));
    const uint32_t s = swz(l);
    tile[s] = make_float2(v.x, v.y);
    tile[s ^ 1u] = make_float2(v.z, v.w);
  }
}

template <int K, int NT>
__device__ __forceinline__ void store_tile(const float2* __restrict__ tile, float2* __restrict__ st,
                                           const TileCtx& t, int tid) {
#pragma unroll 4
  for (int p = tid; p < (1 << (K - 1)); p += NT) {
    const uint32_t l = 2u * p;
    const uint32_t s = swz(l);
    const float2 a = tile[s], b = tile[s ^ 1u];
    *reinterpret_cast<float4*>(st + global_index(t, l)) = make_float4(a.x, a.y, b.x, b.y);
  }
}

// Round geometry.  Thread `tid` owns the 2^R amplitudes whose local index has
// tid's bits deposited on the non-register positions; register value m adds
// the bits of m on the register positions.  In swizzled slot space both parts
// combine by XOR, so the 2^R slots are visited in Gray-code order with one
// v_xor per access: slot(gray(i)) = slot(gray(i-1)) ^ DB[ctz(i)].
template <int K, int R>
__device__ __forceinline__ void round_geometry(uint32_t regmask, int tid, uint32_t (&DB)[R],
                                               uint32_t* T) {
  uint32_t mk = regmask;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    DB[j] = swz(mk & (0u - mk));  // lowest set bit, swizzled
    mk &= mk - 1;
  }
  uint32_t freem = ~regmask & ((1u << K) - 1u);
  uint32_t tl = 0;
#pragma unroll
  for (int j = 0; j < K - R; ++j) {
    const uint32_t low = freem & (0u - freem);
    freem &= freem - 1;
    tl |= ((uint32_t(tid) >> j) & 1u) ? low : 0u;
  }
  *T = swz(tl);
}

template <int R>
__device__ __forceinline__ void round_load(const float2* __restrict__ tile, uint32_t T,
                                           const uint32_t (&DB)[R], float (&ar)[1 << R],
                                           float (&ai)[1 << R]) {
  uint32_t addr = T;
#pragma unroll
  for (int i = 0; i < (1 << R); ++i) {
    if (i) addr ^= DB[__builtin_ctz(i)];
    const float2 v = tile[addr];
    ar[i ^ (i >> 1)] = v.x;
    ai[i ^ (i >> 1)] = v.y;
  }
}

template <int R>
__device__ __forceinline__ void round_store(float2* __restrict__ tile, uint32_t T,
                                            const uint32_t (&DB)[R], const float (&ar)[1 << R],
                                            const float (&ai)[1 << R]) {
  uint32_t addr = T;
#pragma unroll
  for (int i = 0; i < (1 << R); ++i) {
    if (i) addr ^= DB[__builtin_ctz(i)];
    tile[addr] = make_float2(ar[i ^ (i >> 1)], ai[i ^ (i >> 1)]);
  }
}

// Dense two-qubit gate applied directly on the LDS tile (not on the hot path of the
// hardware-efficient ansatz; keeps the register rounds free of 4x4 code).
// pos0/pos1 = local bit of the first/second qubit; matrix index = (b_q0 << 1) | b_q1.
template <int K, int NT>
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

// exp(i*pi*x): the angle is accumulated and range-reduced to [-1, 1] in double, the
// sine/cosine are then evaluated in fp32 (table entries are fp32 anyway).
__device__ __forceinline__ float2 phase_of(double x) {
  const float r = float(x - 2.0 * rint(0.5 * x));
  float sn, cs;
  sincospif(r, &sn, &cs);
  return make_float2(cs, sn);
}

// Diagonal phase tables for one OP_DIAG.  sgn = +1 forward, -1 adjoint (conj).
// Returns through LDS: e_lo[128], e_hi[2^(K-7)], cross[n_cross] = {cos, sin, lmask|par<<31, active/parity}.
template <int K, int NT>
__device__ __forceinline__ void build_diag_tables(const uint32_t* __restrict__ terms, uint32_t n_lo,
                                                  uint32_t n_hi, uint32_t n_cross,
                                                  const double* __restrict__ angles, double sgn,
                                                  const TileCtx& t, float2* e_lo, float2* e_hi,
                                                  float4* cross, int tid) {
  constexpr int NLO = 1 << kLoBits;
  constexpr int NHI = 1 << (K - kLoBits);
  auto entry = [&](uint32_t l, const uint32_t* __restrict__ tp, uint32_t cnt) {
    double ang = 0.0;
    for (uint32_t k = 0; k < cnt; ++k) {
      const uint32_t w0 = uni(tp[k * kDiagTermWords]), nm = uni(tp[k * kDiagTermWords + 1]);
      const uint32_t lm = w0 & 0x7fffffffu;
      const double a = angles[uni(tp[k * kDiagTermWords + 2])];
      bool on;
      if (w0 >> 31) on = (__popc(l & lm) + __popc(t.tile_base & nm)) & 1;
      else on = ((l & lm) == lm) && ((t.tile_base & nm) == nm);
      ang += on ? a : 0.0;
    }
    return phase_of(sgn * ang);
  };
  for (int e = tid; e < NLO; e += NT) e_lo[e] = entry(uint32_t(e), terms, n_lo);
  for (int e = tid; e < NHI; e += NT)
    e_hi[e] = entry(uint32_t(e) << kLoBits, terms + n_lo * kDiagTermWords, n_hi);
  const uint32_t* cp = terms + (n_lo + n_hi) * kDiagTermWords;
  for (uint32_t k = tid; k < n_cross; k += NT) {
    const uint32_t w0 = cp[k * kDiagTermWords], nm = cp[k * kDiagTermWords + 1];
    const float2 ph = phase_of(sgn * angles[cp[k * kDiagTermWords + 2]]);
    const float cs = ph.x, sn = ph.y;
    uint32_t flag;
    if (w0 >> 31) flag = __popc(t.tile_base & nm) & 1;   // parity contributed by nonlocal bits
    else flag = ((t.tile_base & nm) == nm) ? 1u : 0u;       // AND term active on this tile
    cross[k] = make_float4(float(cs), float(sn), __uint_as_float(w0), __uint_as_float(flag));
  }
}

__device__ __forceinline__ float2 diag_phase(uint32_t l, const float2* e_lo, const float2* e_hi,
                                             const float4* cross, uint32_t n_cross) {
  const float2 a = e_lo[l & ((1u << kLoBits) - 1u)], b = e_hi[l >> kLoBits];
  float2 e = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
  for (uint32_t k = 0; k < n_cross; ++k) {
    const float4 ct = cross[k];
    const uint32_t w0 = __float_as_uint(ct.z), flag = __float_as_uint(ct.w);
    const uint32_t lm = w0 & 0x7fffffffu;
    bool on;
    if (w0 >> 31) on = (__popc(l & lm) + flag) & 1;
    else on = flag && ((l & lm) == lm);
    if (on) e = make_float2(e.x * ct.x - e.y * ct.y, e.x * ct.y + e.y * ct.x);
  }
  return e;
}

__device__ __forceinline__ uint32_t basis_index(const int8_t* __restrict__ row, int n_user) {
  uint32_t idx = 0;
  for (int q = 0; q < n_user; ++q) idx |= (row[q] ? 1u : 0u) << (n_user - 1 - q);
  return idx;
}

}  // namespace

// ================================================================================
// Forward pass kernel
// ================================================================================
template <int K, int R>
__global__ __launch_bounds__(1 << (K - R), fwd_min_waves(K, R)) void pass_fwd_kernel(
    PassArgs a, float2* __restrict__ psi, const int8_t* __restrict__ bits, int n_user,
    const uint32_t* __restrict__ prog_base, const uint32_t* __restrict__ tables,
    const float* __restrict__ coef, const double* __restrict__ angles, float* __restrict__ out,
    uint32_t state0) {
  constexpr int NT = 1 << (K - R);
  constexpr int NR = 1 << R;
  constexpr int NHI = 1 << (K - kLoBits);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tile = reinterpret_cast<float2*>(smem);
  float2* e_lo = tile + (1 << K);
  float2* e_hi = e_lo + (1 << kLoBits);
  float4* cross = reinterpret_cast<float4*>(e_hi + NHI);
  float* red = reinterpret_cast<float*>(cross + kMaxCrossTerms);

  const int tid = threadIdx.x;
  const uint32_t tile_id = blockIdx.x & ((1u << a.n_nonlocal) - 1u);
  const uint32_t s_local = blockIdx.x >> a.n_nonlocal;
  TileCtx t;
  {
    uint32_t tb = 0;
    for (uint32_t i = 0; i < a.n_nonlocal; ++i) tb |= ((tile_id >> i) & 1u) << a.nonlocal_pos[i];
    t.tile_base = tb;
    t.c = a.c;
    t.cmask = (1u << a.c) - 1u;
    t.spread = tables + a.spread_off;
  }
  float2* st = psi + (size_t(s_local) << a.n);

  if (a.flags & PASS_INIT_BASIS) {
    for (int l = tid; l < (1 << K); l += NT) tile[l] = make_float2(0.f, 0.f);
    __syncthreads();
    if (tid == 0) {
      const uint32_t idx = basis_index(bits + size_t(state0 + s_local) * n_user, n_user);
      uint32_t nl_mask = 0;
      for (uint32_t i = 0; i < a.n_nonlocal; ++i) nl_mask |= 1u << a.nonlocal_pos[i];
      if ((idx & nl_mask) == t.tile_base) {
        uint32_t l = 0;
        for (int i = 0; i < K; ++i) l |= ((idx >> a.local_pos[i]) & 1u) << i;
        tile[swz(l)] = make_float2(1.f, 0.f);
      }
    }
  } else {
    load_tile<K, NT>(tile, st, t, tid);
  }
  for (int i = tid; i < kMaxOps; i += NT) red[i] = 0.f;
  __syncthreads();

  const uint32_t* prog = prog_base + a.prog_off;
  uint32_t pc = 0;
  for (;;) {
    const uint32_t w0 = uni(prog[pc]);
    const uint32_t opc = w0 & 0xffu;
    if (opc == OP_END) break;
    if (opc == OP_ROUND) {
      const uint32_t n_micro = w0 >> 8;
      const uint32_t regmask = uni(prog[pc + 1]);
      uint32_t DB[R], T;
      round_geometry<K, R>(regmask, tid, DB, &T);
      float ar[NR], ai[NR];
      round_load<R>(tile, T, DB, ar, ai);
      const uint32_t* mp = prog + pc + 2;
      for (uint32_t i = 0; i < n_micro; ++i, mp += kMicroWords) {
        const uint32_t mw = uni(mp[0]);
        const float* cf = coef + uni(mp[1]);
        const uint32_t mop = mw & 0xffu, rb0 = (mw >> 8) & 15u;
        if (mop == MOP_X) {
          const float c = cf[0], s = cf[1];
          QHBM_DISPATCH_RB(R, rb0, (apply_x<R, RB>(ar, ai, c, s)));
        } else if (mop == MOP_Y) {
          const float c = cf[0], s = cf[1];
          QHBM_DISPATCH_RB(R, rb0, (apply_y<R, RB>(ar, ai, c, s)));
        } else {
          QHBM_DISPATCH_RB(R, rb0, (apply_mat1<R, RB>(ar, ai, cf)));
        }
      }
      round_store<R>(tile, T, DB, ar, ai);
      __syncthreads();
      pc += 2 + n_micro * kMicroWords;
    } else if (opc == OP_GATE2) {
      const uint32_t pw = uni(prog[pc + 1]);
      const uint32_t pos0 = pw & 0xffu, pos1 = (pw >> 8) & 0xffu;
      const float* cf = coef + uni(prog[pc + 2]);
      for (uint32_t q = tid; q < (1u << (K - 2)); q += NT) {
        uint32_t ix[4];
        quad_indices<K, NT>(q, pos0, pos1, ix);
        float2 x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = tile[ix[j]];
        mat4_apply(cf, x, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[ix[j]] = y[j];
      }
      __syncthreads();
      pc += kGate2Words;
    } else if (opc == OP_DIAG) {
      const uint32_t cw = uni(prog[pc + 1]);
      const uint32_t n_lo = cw & 1023u, n_hi = (cw >> 10) & 1023u, n_cross = cw >> 20;
      build_diag_tables<K, NT>(prog + pc + 2, n_lo, n_hi, n_cross, angles, 1.0, t, e_lo, e_hi,
                               cross, tid);
      __syncthreads();
#pragma unroll 4
      for (int i = 0; i < NR; ++i) {
        const uint32_t l = uint32_t(i) * NT + tid;
        const float2 e = diag_phase(l, e_lo, e_hi, cross, n_cross);
        const uint32_t s = swz(l);
        const float2 v = tile[s];
        tile[s] = make_float2(v.x * e.x - v.y * e.y, v.x * e.y + v.y * e.x);
      }
      __syncthreads();
      pc += 2 + (n_lo + n_hi + n_cross) * kDiagTermWords;
    } else {  // OP_MEASURE
      const uint32_t n_groups = w0 >> 8;
      pc += 1;
      float acc = 0.f;
      uint32_t cur_op = 0xffffffffu;
      auto flush = [&]() {
        if (cur_op != 0xffffffffu) {
          const float v = wave_sum(acc);
          if ((tid & 63) == 0) atomicAdd(&red[cur_op], v);
        }
        acc = 0.f;
      };
      for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t xl = uni(prog[pc]), n_terms = uni(prog[pc + 1]);
        pc += 2;
        float wr[NR], wi[NR];
        const uint32_t xs = swz(xl);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const uint32_t s = swz(uint32_t(i) * NT + tid);
          const float2 p = tile[s];
          const float2 q = tile[s ^ xs];  // psi[l ^ x]
          // w = conj(psi[l^x]) * psi[l]
          wr[i] = q.x * p.x + q.y * p.y;
          wi[i] = q.x * p.y - q.y * p.x;
        }
        for (uint32_t k = 0; k < n_terms; ++k, pc += kMeasTermWords) {
          const uint32_t zl = uni(prog[pc]), zn = uni(prog[pc + 1]);
          const float cf = __uint_as_float(uni(prog[pc + 2]));
          const uint32_t ow = uni(prog[pc + 3]);
          const uint32_t op = ow & 0xffffffu, ny = ow >> 24;
          if (op != cur_op) { flush(); cur_op = op; }
          // Re( i^ny * (-1)^{popc(l & z)} * w ):  ny=0: wr, 1: -wi, 2: -wr, 3: wi
          float sfac = (ny == 1 || ny == 2) ? -cf : cf;
          if (__popc(t.tile_base & zn) & 1) sfac = -sfac;
          if (__popc(uint32_t(tid) & zl) & 1) sfac = -sfac;
          const uint32_t zhi = zl >> (K - R);  // bits of l above the thread index
          float sum = 0.f;
          if (ny & 1) {
#pragma unroll
            for (int i = 0; i < NR; ++i) sum += (__builtin_popcount(uint32_t(i) & zhi) & 1) ? -wi[i] : wi[i];
          } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) sum += (__builtin_popcount(uint32_t(i) & zhi) & 1) ? -wr[i] : wr[i];
          }
          acc = fmaf(sfac, sum, acc);
        }
      }
      flush();
      __syncthreads();
    }
  }

  if (a.n_ops) {
    __syncthreads();
    for (uint32_t i = tid; i < a.n_ops; i += NT) {
      const float v = red[i];
      if (v != 0.f) atomicAdd(&out[size_t(state0 + s_local) * a.n_ops + i], v);
    }
  }
  if (a.flags & PASS_STORE) store_tile<K, NT>(tile, st, t, tid);
}

// ================================================================================
// Adjoint pass kernel: tile pair (psi, lambda); program already in reverse order.
// For each parametrised gate:  dE/dt = -2*pi * Im <lam| A |psi>  with psi, lam taken
// AFTER the gate and A = sum_k e_k P_k, then both are multiplied by U^dagger.
// ================================================================================
template <int K>
__global__ __launch_bounds__(1 << (K - 4), adj_min_waves(K)) void pass_adj_kernel(
    PassArgs a, float2* __restrict__ psi, float2* __restrict__ lam,
    const uint32_t* __restrict__ prog_base, const uint32_t* __restrict__ tables,
    const float* __restrict__ coef, const double* __restrict__ angles,
    float* __restrict__ state_grad /*[U, n_slots_total]*/, uint32_t n_slots_total,
    uint32_t state0) {
  constexpr int R = 4;
  constexpr int NT = 1 << (K - R);
  constexpr int NR = 1 << R;
  constexpr int NHI = 1 << (K - kLoBits);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tp = reinterpret_cast<float2*>(smem);
  float2* tl = tp + (1 << K);
  float2* e_lo = tl + (1 << K);
  float2* e_hi = e_lo + (1 << kLoBits);
  float4* cross = reinterpret_cast<float4*>(e_hi + NHI);
  float* sacc = reinterpret_cast<float*>(cross + kMaxCrossTerms);  // [kMaxSlotsPerPass]

  const int tid = threadIdx.x;
  const uint32_t tile_id = blockIdx.x & ((1u << a.n_nonlocal) - 1u);
  const uint32_t s_local = blockIdx.x >> a.n_nonlocal;
  TileCtx t;
  {
    uint32_t tb = 0;
    for (uint32_t i = 0; i < a.n_nonlocal; ++i) tb |= ((tile_id >> i) & 1u) << a.nonlocal_pos[i];
    t.tile_base = tb;
    t.c = a.c;
    t.cmask = (1u << a.c) - 1u;
    t.spread = tables + a.spread_off;
  }
  float2* sp = psi + (size_t(s_local) << a.n);
  float2* sl = lam + (size_t(s_local) << a.n);
  load_tile<K, NT>(tp, sp, t, tid);
  load_tile<K, NT>(tl, sl, t, tid);
  for (uint32_t i = tid; i < a.n_slots; i += NT) sacc[i] = 0.f;
  __syncthreads();

  auto add_slot = [&](uint32_t slot, float v) {
    v = wave_sum(v);
    if ((tid & 63) == 0) atomicAdd(&sacc[slot - a.slot_base], v);
  };

  const uint32_t* prog = prog_base + a.prog_off;
  uint32_t pc = 0;
  for (;;) {
    const uint32_t w0 = uni(prog[pc]);
    const uint32_t opc = w0 & 0xffu;
    if (opc == OP_END) break;
    if (opc == OP_ROUND) {
      const uint32_t n_micro = w0 >> 8;
      const uint32_t regmask = uni(prog[pc + 1]);
      uint32_t DB[R], T;
      round_geometry<K, R>(regmask, tid, DB, &T);
      float pr[NR], pi[NR], lr[NR], li[NR];
      round_load<R>(tp, T, DB, pr, pi);
      round_load<R>(tl, T, DB, lr, li);
      const uint32_t* mp = prog + pc + 2;
      for (uint32_t i = 0; i < n_micro; ++i, mp += kMicroWords) {
        const uint32_t mw = uni(mp[0]);
        const float* cf = coef + uni(mp[1]);
        const uint32_t slot = uni(mp[2]);
        const uint32_t mop = mw & 0xffu, rb0 = (mw >> 8) & 15u;
        const bool want = slot != 0xffffffffu;
        float g = 0.f;
        if (mop == MOP_X) {
          const float c = cf[0], s = -cf[1];  // U^dagger = c*I + i*s*X
          QHBM_DISPATCH_RB(R, rb0, (g = want ? kPi * im_lam_x_psi<R, RB>(pr, pi, lr, li) : 0.f,
                                   apply_x<R, RB>(pr, pi, c, s), apply_x<R, RB>(lr, li, c, s)));
        } else if (mop == MOP_Y) {
          const float c = cf[0], s = -cf[1];
          QHBM_DISPATCH_RB(R, rb0, (g = want ? kPi * im_lam_y_psi<R, RB>(pr, pi, lr, li) : 0.f,
                                   apply_y<R, RB>(pr, pi, c, s), apply_y<R, RB>(lr, li, c, s)));
        } else {
          QHBM_DISPATCH_RB(R, rb0, (g = want ? im_lam_g1_psi<R, RB>(pr, pi, lr, li, cf + 8) : 0.f,
                                   apply_mat1<R, RB>(pr, pi, cf), apply_mat1<R, RB>(lr, li, cf)));
        }
        if (want) add_slot(slot, g);
      }
      round_store<R>(tp, T, DB, pr, pi);
      round_store<R>(tl, T, DB, lr, li);
      __syncthreads();
      pc += 2 + n_micro * kMicroWords;
    } else if (opc == OP_GATE2) {
      const uint32_t pw = uni(prog[pc + 1]);
      const uint32_t pos0 = pw & 0xffu, pos1 = (pw >> 8) & 0xffu;
      const float* cf = coef + uni(prog[pc + 2]);  // U^dagger (32 floats) then generator (32 floats)
      const uint32_t slot = uni(prog[pc + 3]);
      float gacc = 0.f;
      for (uint32_t q = tid; q < (1u << (K - 2)); q += NT) {
        uint32_t ix[4];
        quad_indices<K, NT>(q, pos0, pos1, ix);
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
      if (slot != 0xffffffffu) add_slot(slot, gacc);
      __syncthreads();
      pc += kGate2Words;
    } else {  // OP_DIAG
      const uint32_t cw = uni(prog[pc + 1]);
      const uint32_t n_lo = cw & 1023u, n_hi = (cw >> 10) & 1023u, n_cross = cw >> 20;
      const uint32_t n_terms = n_lo + n_hi + n_cross;
      const uint32_t* terms = prog + pc + 2;
      build_diag_tables<K, NT>(terms, n_lo, n_hi, n_cross, angles, -1.0, t, e_lo, e_hi, cross, tid);
      // w_i = Im(conj(lam) psi) at l = i*NT + tid (invariant under the diagonal itself)
      float w[NR];
      float wtot = 0.f;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const uint32_t s = swz(uint32_t(i) * NT + tid);
        const float2 p = tp[s], q = tl[s];
        w[i] = q.x * p.y - q.y * p.x;
        wtot += w[i];
      }
      for (uint32_t k = 0; k < n_terms; ++k) {
        const uint32_t tw = uni(terms[k * kDiagTermWords]), nm = uni(terms[k * kDiagTermWords + 1]);
        const uint32_t slot = uni(terms[k * kDiagTermWords + 3]);
        if (slot == 0xffffffffu) continue;
        const uint32_t lm = tw & 0x7fffffffu;
        const uint32_t lm_t = lm & (NT - 1u), lm_h = lm >> (K - R);
        float v;
        if (tw >> 31) {  // parity term: sum_l w(l) * parity(idx & mask)
          float asum = 0.f;
#pragma unroll
          for (int i = 0; i < NR; ++i) asum += (__builtin_popcount(uint32_t(i) & lm_h) & 1) ? w[i] : 0.f;
          const bool pt = (__popc(uint32_t(tid) & lm_t) + __popc(t.tile_base & nm)) & 1;
          v = pt ? (wtot - asum) : asum;
        } else {  // AND term
          float asum = 0.f;
#pragma unroll
          for (int i = 0; i < NR; ++i) asum += ((uint32_t(i) & lm_h) == lm_h) ? w[i] : 0.f;
          const bool on = ((uint32_t(tid) & lm_t) == lm_t) && ((t.tile_base & nm) == nm);
          v = on ? asum : 0.f;
        }
        add_slot(slot, -2.f * kPi * v);
      }
      __syncthreads();
#pragma unroll 4
      for (int i = 0; i < NR; ++i) {
        const uint32_t l = uint32_t(i) * NT + tid;
        const float2 e = diag_phase(l, e_lo, e_hi, cross, n_cross);
        const uint32_t s = swz(l);
        const float2 v = tp[s], u = tl[s];
        tp[s] = make_float2(v.x * e.x - v.y * e.y, v.x * e.y + v.y * e.x);
        tl[s] = make_float2(u.x * e.x - u.y * e.y, u.x * e.y + u.y * e.x);
      }
      __syncthreads();
      pc += 2 + n_terms * kDiagTermWords;
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < a.n_slots; i += NT) {
    const float v = sacc[i];
    if (v != 0.f) atomicAdd(&state_grad[size_t(state0 + s_local) * n_slots_total + a.slot_base + i], v);
  }
  if (a.flags & PASS_STORE) {
    store_tile<K, NT>(tp, sp, t, tid);
    store_tile<K, NT>(tl, sl, t, tid);
  }
}

// ================================================================================
// lambda = sum_k upstream[s, op_k] * c_k * P_k psi      (one thread per amplitude)
// (P psi)[j] = i^ny (-1)^{popc((j^x) & z)} psi[j ^ x]
// ================================================================================
__global__ __launch_bounds__(256) void apply_observable_kernel(
    const float2* __restrict__ psi, float2* __restrict__ lam, uint32_t n, const DevTerm* __restrict__ terms,
    uint32_t n_terms, const float* __restrict__ upstream, uint32_t n_ops, uint32_t state0) {
  const uint32_t s_local = blockIdx.y;
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  const float2* ps = psi + (size_t(s_local) << n);
  const float* up = upstream + size_t(state0 + s_local) * n_ops;
  float ar = 0.f, ai = 0.f;
  for (uint32_t k = 0; k < n_terms; ++k) {
    const DevTerm tm = terms[k];
    const float w = up[tm.op] * tm.coeff;
    const uint32_t src = j ^ tm.x;
    const float2 v = ps[src];
    const float sg = (__popc(src & tm.z) & 1) ? -w : w;
    switch (tm.ny & 3) {
      case 0: ar += sg * v.x; ai += sg * v.y; break;
      case 1: ar -= sg * v.y; ai += sg * v.x; break;
      case 2: ar -= sg * v.x; ai -= sg * v.y; break;
      default: ar += sg * v.y; ai -= sg * v.x; break;
    }
  }
  lam[(size_t(s_local) << n) + j] = make_float2(ar, ai);
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

__global__ void prep_coefs_kernel(const CoefJob* __restrict__ jobs, int n_jobs,
                                  const float* __restrict__ params, float* __restrict__ coef,
                                  double* __restrict__ angles, int shift_gate, double shift) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  const CoefJob jb = jobs[j];
  double t = double(jb.offset);
  if (jb.param_idx >= 0) t += double(jb.scalar) * double(params[jb.param_idx]);
  if (jb.gate == shift_gate) t += shift;
  if (jb.mop == 0) { angles[jb.out_off] = t; return; }
  float* o = coef + jb.out_off;
  double sh, ch;  // sin, cos of pi*t/2
  sincospi(0.5 * t, &sh, &ch);
  if (jb.mop == MOP_X || jb.mop == MOP_Y) {
    o[0] = float(ch);
    o[1] = float(sh);
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

// grad[p] += weight * sum_{s,k} upstream[s,k] * (vals_plus - vals_minus)[s,k]   (parameter shift)
__global__ __launch_bounds__(256) void shift_accumulate_kernel(
    const float* __restrict__ vp, const float* __restrict__ vm, const float* __restrict__ upstream,
    uint32_t count, float weight, float* __restrict__ grad_p) {
  __shared__ double part[256];
  double acc = 0.0;
  for (uint32_t i = threadIdx.x; i < count; i += 256) acc += double(upstream[i]) * (double(vp[i]) - double(vm[i]));
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (int(threadIdx.x) < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *grad_p += weight * float(part[0]);
}

// ================================================================================
// Host-side launchers
// ================================================================================
size_t fwd_lds_bytes(int K) {
  return (size_t(1) << K) * 8 + (size_t(1) << kLoBits) * 8 + (size_t(1) << (K - kLoBits)) * 8 +
         size_t(kMaxCrossTerms) * 16 + size_t(kMaxOps) * 4;
}
size_t adj_lds_bytes(int K) {
  return (size_t(2) << K) * 8 + (size_t(1) << kLoBits) * 8 + (size_t(1) << (K - kLoBits)) * 8 +
         size_t(kMaxCrossTerms) * 16 + size_t(kMaxSlotsPerPass) * 4;
}

template <int K, int R>
static hipError_t launch_fwd_t(const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits,
                               int n_user, const uint32_t* prog, const uint32_t* tables,
                               const float* coef, const double* angles, float* out, uint32_t state0,
                               hipStream_t stream) {
  const size_t lds = fwd_lds_bytes(K);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_fwd_kernel<K, R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const uint32_t grid = n_states << a.n_nonlocal;
  hipLaunchKernelGGL((pass_fwd_kernel<K, R>), dim3(grid), dim3(1 << (K - R)), lds, stream, a, psi, bits,
                     n_user, prog, tables, coef, angles, out, state0);
  return hipGetLastError();
}

hipError_t launch_pass_fwd(int K, int R, const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits,
                           int n_user, const uint32_t* prog, const uint32_t* tables, const float* coef,
                           const double* angles, float* out, uint32_t state0, hipStream_t stream) {
#define QHBM_FWD_CASE(K_, R_)                                                                          \
  if (K == K_ && R == R_)                                                                              \
    return launch_fwd_t<K_, R_>(a, n_states, psi, bits, n_user, prog, tables, coef, angles, out, state0, stream);
  QHBM_FWD_CASE(10, 4)
  QHBM_FWD_CASE(11, 4)
  QHBM_FWD_CASE(12, 4)
  QHBM_FWD_CASE(12, 5)
  QHBM_FWD_CASE(13, 4)
  QHBM_FWD_CASE(13, 5)
  QHBM_FWD_CASE(14, 4)
  QHBM_FWD_CASE(14, 5)
#undef QHBM_FWD_CASE
  return hipErrorInvalidValue;
}

template <int K>
static hipError_t launch_adj_t(const PassArgs& a, uint32_t n_states, float2* psi, float2* lam,
                               const uint32_t* prog, const uint32_t* tables, const float* coef,
                               const double* angles, float* state_grad, uint32_t n_slots_total,
                               uint32_t state0, hipStream_t stream) {
  const size_t lds = adj_lds_bytes(K);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_adj_kernel<K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const uint32_t grid = n_states << a.n_nonlocal;
  hipLaunchKernelGGL((pass_adj_kernel<K>), dim3(grid), dim3(1 << (K - 4)), lds, stream, a, psi, lam, prog,
                     tables, coef, angles, state_grad, n_slots_total, state0);
  return hipGetLastError();
}

hipError_t launch_pass_adj(int K, const PassArgs& a, uint32_t n_states, float2* psi, float2* lam,
                           const uint32_t* prog, const uint32_t* tables, const float* coef,
                           const double* angles, float* state_grad, uint32_t n_slots_total,
                           uint32_t state0, hipStream_t stream) {
  switch (K) {
    case 10: return launch_adj_t<10>(a, n_states, psi, lam, prog, tables, coef, angles, state_grad, n_slots_total, state0, stream);
    case 11: return launch_adj_t<11>(a, n_states, psi, lam, prog, tables, coef, angles, state_grad, n_slots_total, state0, stream);
    case 12: return launch_adj_t<12>(a, n_states, psi, lam, prog, tables, coef, angles, state_grad, n_slots_total, state0, stream);
    case 13: return launch_adj_t<13>(a, n_states, psi, lam, prog, tables, coef, angles, state_grad, n_slots_total, state0, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_apply_observable(const float2* psi, float2* lam, uint32_t n, uint32_t n_states,
                                   const DevTerm* terms, uint32_t n_terms, const float* upstream,
                                   uint32_t n_ops, uint32_t state0, hipStream_t stream) {
  const uint32_t blocks = (1u << n) / 256u;
  hipLaunchKernelGGL(apply_observable_kernel, dim3(blocks, n_states), dim3(256), 0, stream, psi, lam, n,
                     terms, n_terms, upstream, n_ops, state0);
  return hipGetLastError();
}

hipError_t launch_prep_coefs(const CoefJob* jobs, int n_jobs, const float* params, float* coef,
                             double* angles, int shift_gate, double shift, hipStream_t stream) {
  if (n_jobs == 0) return hipSuccess;
  hipLaunchKernelGGL(prep_coefs_kernel, dim3((n_jobs + 127) / 128), dim3(128), 0, stream, jobs, n_jobs,
                     params, coef, angles, shift_gate, shift);
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

hipError_t launch_shift_accumulate(const float* vp, const float* vm, const float* upstream,
                                   uint32_t count, float weight, float* grad_p, hipStream_t stream) {
  hipLaunchKernelGGL(shift_accumulate_kernel, dim3(1), dim3(256), 0, stream, vp, vm, upstream, count,
                     weight, grad_p);
  return hipGetLastError();
}

}  // namespace qhbm
