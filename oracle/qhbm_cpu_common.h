/*
 * qhbm_cpu_common.h -- what the two CPU restatements (qhbm_cpu.c: the gate-by-gate CHECKER; qhbm_cpu_diag.c: the timed
 * baseline with diagonal merging) share: the gate record, cirq's EigenGate matrices, Pauli phases, the threading policy.
 * TEST INFRASTRUCTURE, NOT THE PRODUCT (see qhbm_cpu.c).
 */
#ifndef QHBM_CPU_COMMON_H_
#define QHBM_CPU_COMMON_H_
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int32_t kind, q0, q1, param_idx;
  float scalar, offset;
} qo_gate;

typedef float complex cf;
typedef double complex cd;

/* Threads: across states (TFQ's policy for batches of small circuits) or INSIDE a state (choose_threading below):
 * the same gate-by-gate arithmetic with the sweep over the amplitudes split over the team (sums of a sweep are
 * double-precision reductions, so only their order changes). */
static int g_inner = 0;

enum { G_I = 0, G_X, G_Y, G_Z, G_H, G_CZ, G_CNOT, G_SWAP, G_ISWAP, G_XX, G_YY, G_ZZ };

static int two_qubit(int kind) { return kind >= G_CZ; }

/* Base involution of a kind as a dense matrix (dim 2 or 4, row-major; first
 * qubit = high index bit). */
static void base_matrix(int kind, cd* g) {
  const double r = 0.70710678118654752440;
  int dim = two_qubit(kind) ? 4 : 2;
  memset(g, 0, sizeof(cd) * dim * dim);
  switch (kind) {
    case G_I: g[0] = g[3] = 1; break;
    case G_X: g[1] = g[2] = 1; break;
    case G_Y: g[1] = -I; g[2] = I; break;
    case G_Z: g[0] = 1; g[3] = -1; break;
    case G_H: g[0] = r; g[1] = r; g[2] = r; g[3] = -r; break;
    case G_CZ: g[0] = g[5] = g[10] = 1; g[15] = -1; break;
    case G_CNOT: g[0] = g[5] = 1; g[11] = g[14] = 1; break;
    case G_SWAP: g[0] = g[15] = 1; g[6] = g[9] = 1; break;
    case G_XX: g[3] = g[6] = g[9] = g[12] = 1; break;
    case G_YY: g[3] = -1; g[6] = 1; g[9] = 1; g[12] = -1; break;
    case G_ZZ: g[0] = 1; g[5] = -1; g[10] = -1; g[15] = 1; break;
    default: break;
  }
}

/* u = G**t, du = d/dt G**t (dim x dim, row-major). */
static void gate_matrices(int kind, double t, cd* u, cd* du) {
  int dim = two_qubit(kind) ? 4 : 2;
  if (kind == G_ISWAP) {
    /* eigen-exponents 0 (|00>,|11>), +1/2 ((|01>+|10>)/sqrt2), -1/2 ((|01>-|10>)/sqrt2) */
    cd ep = cexp(I * M_PI * t * 0.5), em = cexp(-I * M_PI * t * 0.5);
    cd dp = (I * M_PI * 0.5) * ep, dm = (-I * M_PI * 0.5) * em;
    memset(u, 0, sizeof(cd) * 16);
    memset(du, 0, sizeof(cd) * 16);
    u[0] = u[15] = 1;
    u[5] = u[10] = 0.5 * (ep + em);
    u[6] = u[9] = 0.5 * (ep - em);
    du[5] = du[10] = 0.5 * (dp + dm);
    du[6] = du[9] = 0.5 * (dp - dm);
    return;
  }
  cd g[16];
  base_matrix(kind, g);
  if (kind == G_I) {
    for (int i = 0; i < 4; ++i) { u[i] = g[i]; du[i] = 0; }
    return;
  }
  cd ph = cexp(I * M_PI * t), dph = (I * M_PI) * ph;
  for (int i = 0; i < dim; ++i)
    for (int j = 0; j < dim; ++j) {
      cd id = (i == j) ? 1.0 : 0.0;
      /* (I+G)/2 + ph (I-G)/2 */
      u[i * dim + j] = 0.5 * (id + g[i * dim + j]) + ph * 0.5 * (id - g[i * dim + j]);
      du[i * dim + j] = dph * 0.5 * (id - g[i * dim + j]);
    }
}

static double exponent_of(const qo_gate* g, const float* params) {
  double t = g->offset;
  if (g->param_idx >= 0) t += (double)g->scalar * (double)params[g->param_idx];
  return t;
}

static uint64_t to_index_mask(uint64_t qmask, int n) {
  uint64_t m = 0;
  for (int q = 0; q < n; ++q) if (qmask >> q & 1) m |= (uint64_t)1 << (n - 1 - q);
  return m;
}

/* phase of (P psi)[j] = i^ny (-1)^{popc(src & z)} psi[src], src = j ^ x */
static cf pauli_phase(int ny, uint64_t src, uint64_t z) {
  static const cf ipow[4] = {1, I, -1, -I};
  cf p = ipow[ny & 3];
  return (__builtin_popcountll(src & z) & 1) ? -p : p;
}

/* The team the caller asked for (n_threads > 0) or the machine offers -- remembered, because a call that threads INSIDE
 * a state shrinks OpenMP's team for itself.  Threads go ACROSS states (one state per thread) unless the call holds very few
 * large states: inside a state the speed-up is modest (a sweep of 2^20 amplitudes gains ~5 x from 8 threads and LOSES from
 * 32: forks and spinning; measured on the GPU box's 256 hardware threads, where 19 qubits inside a state on the whole team
 * ran 400 x slower than 17 qubits across states), so 40 states of 20 qubits are faster across 40 threads.  Inside a state: from
 * 26 qubits always (TFQ's policy), from 22 qubits for at most 8 states, from 18 qubits for at most 2; one thread per 2^17
 * amplitudes, at most 64 (memory bandwidth is spent by then). */
static int g_full_team = 0;

static int team_size(void) {
#ifdef _OPENMP
  if (g_full_team <= 0) g_full_team = omp_get_max_threads();
  return g_full_team;
#else
  return 1;
#endif
}

static void choose_threading(int n, int U, int n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) g_full_team = n_threads;
#endif
  const int team = team_size();
  int inner = (int)(((size_t)1 << n) >> 17);
  if (inner > 64) inner = 64;
  if (inner > team) inner = team;
  g_inner = inner > 1 && U < team && (n >= 26 || (n >= 22 && U <= 8) || (n >= 18 && U <= 2));
#ifdef _OPENMP
  omp_set_num_threads(g_inner ? inner : team);
#endif
}

static void apply1(cf* psi, int n, int bit, const cf* m) {
  const size_t dim = (size_t)1 << n, half = dim >> 1, st = (size_t)1 << bit;
  if (!g_inner) {
    for (size_t base = 0; base < dim; base += 2 * st)
      for (size_t k = base; k < base + st; ++k) {
        cf a = psi[k], b = psi[k + st];
        psi[k] = m[0] * a + m[1] * b;
        psi[k + st] = m[2] * a + m[3] * b;
      }
    return;
  }
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < half; ++i) { /* the same pairs, dealt to the team */
    const size_t k = ((i >> bit) << (bit + 1)) | (i & (st - 1));
    cf a = psi[k], b = psi[k + st];
    psi[k] = m[0] * a + m[1] * b;
    psi[k + st] = m[2] * a + m[3] * b;
  }
}

static void apply2(cf* psi, int n, int bit_hi_q0, int bit_q1, const cf* m) {
  /* matrix index = (b_q0 << 1) | b_q1 */
  const size_t dim = (size_t)1 << n, s0 = (size_t)1 << bit_hi_q0, s1 = (size_t)1 << bit_q1;
#pragma omp parallel for schedule(static) if (g_inner)
  for (size_t k = 0; k < dim; ++k) {
    if (k & (s0 | s1)) continue;
    cf x[4] = {psi[k], psi[k | s1], psi[k | s0], psi[k | s0 | s1]}, y[4];
    for (int i = 0; i < 4; ++i) y[i] = m[i * 4] * x[0] + m[i * 4 + 1] * x[1] + m[i * 4 + 2] * x[2] + m[i * 4 + 3] * x[3];
    psi[k] = y[0]; psi[k | s1] = y[1]; psi[k | s0] = y[2]; psi[k | s0 | s1] = y[3];
  }
}

static void apply_gate_matrix(cf* psi, int n, const qo_gate* g, const cd* u, int dagger) {
  int dim = two_qubit(g->kind) ? 4 : 2;
  cf m[16];
  for (int i = 0; i < dim; ++i)
    for (int j = 0; j < dim; ++j) m[i * dim + j] = dagger ? (cf)conj(u[j * dim + i]) : (cf)u[i * dim + j];
  if (dim == 2) apply1(psi, n, n - 1 - g->q0, m);
  else apply2(psi, n, n - 1 - g->q0, n - 1 - g->q1, m);
}

/* 2 Re <lam| dU |psi> without materialising dU psi. */
static double inner_du(const cf* lam, const cf* psi, int n, const qo_gate* g, const cd* du) {
  const size_t dim = (size_t)1 << n;
  double acc = 0.0;
  if (!two_qubit(g->kind)) {
    const size_t st = (size_t)1 << (n - 1 - g->q0);
    cf m[4];
    for (int i = 0; i < 4; ++i) m[i] = (cf)du[i];
    const int bit = n - 1 - g->q0;
    const size_t half = dim >> 1;
    if (!g_inner) {
      for (size_t base = 0; base < dim; base += 2 * st)
        for (size_t k = base; k < base + st; ++k) {
          cf a = psi[k], b = psi[k + st];
          acc += creal(conjf(lam[k]) * (m[0] * a + m[1] * b) + conjf(lam[k + st]) * (m[2] * a + m[3] * b));
        }
    } else {
#pragma omp parallel for schedule(static) reduction(+ : acc)
      for (size_t i = 0; i < half; ++i) {
        const size_t k = ((i >> bit) << (bit + 1)) | (i & (st - 1));
        cf a = psi[k], b = psi[k + st];
        acc += creal(conjf(lam[k]) * (m[0] * a + m[1] * b) + conjf(lam[k + st]) * (m[2] * a + m[3] * b));
      }
    }
  } else {
    const size_t s0 = (size_t)1 << (n - 1 - g->q0), s1 = (size_t)1 << (n - 1 - g->q1);
    cf m[16];
    for (int i = 0; i < 16; ++i) m[i] = (cf)du[i];
#pragma omp parallel for schedule(static) reduction(+ : acc) if (g_inner)
    for (size_t k = 0; k < dim; ++k) {
      if (k & (s0 | s1)) continue;
      const size_t ix[4] = {k, k | s1, k | s0, k | s0 | s1};
      for (int i = 0; i < 4; ++i) {
        cf y = 0;
        for (int j = 0; j < 4; ++j) y += m[i * 4 + j] * psi[ix[j]];
        acc += creal(conjf(lam[ix[i]]) * y);
      }
    }
  }
  return 2.0 * acc;
}

#endif  /* QHBM_CPU_COMMON_H_ */
