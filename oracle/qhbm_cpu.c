/*
 * qhbm_cpu.c -- fp32 CPU restatement (C + OpenMP) of the expectation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  It is the C twin of
 * oracle/qhbm_oracle.py and serves (a) as a second, independently written
 * checker and (b) as the timed CPU baseline of bench.py (`cpu_baseline.kind`
 * = "port").  The product path (qhbm-library_amd/) never links or calls it.
 *
 * What it restates: the per-circuit work of tfq.layers.Expectation as invoked
 * at /root/reference/qhbmlib/inference/qnn.py:134-138 -- fp32 statevector
 * simulation from the basis state injected by
 * /root/reference/qhbmlib/models/circuit.py:129-136, gate by gate (no fusion),
 * Pauli-sum expectation, and the adjoint backward that TFQ uses for noiseless
 * expectation (qnn.py:90-99).  Threads run across states, which is TFQ's
 * policy for batches of small circuits [SURVEY.md 8d].  Gate matrices follow
 * cirq 0.14.1's EigenGate convention G**t = sum_k exp(i pi t e_k) P_k
 * (same table as oracle/qhbm_oracle.py::_eigen_components).
 *
 * Parity pin: checked against oracle/qhbm_oracle.py (itself pinned to the
 * reference's closed-form KATs) in tests/test_oracle_c.py.
 */
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

/* Threads: across states (TFQ's policy for batches of small circuits), or -- when a call holds fewer states than
 * threads and the state is large (qo_expectation*: U < threads / 2 and n >= 18), and always from 26 qubits (TFQ's
 * policy for large circuits [SURVEY.md 8d]) -- INSIDE a state: the same gate-by-gate arithmetic with the sweep over the
 * amplitudes split over the team (sums of a sweep are double-precision reductions, so only their order changes). */
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

static void simulate(cf* psi, int n, int n_gates, const qo_gate* gates, const float* params, const int8_t* bits) {
  const size_t dim = (size_t)1 << n;
  memset(psi, 0, dim * sizeof(cf));
  size_t idx = 0;
  for (int q = 0; q < n; ++q) if (bits[q]) idx |= (size_t)1 << (n - 1 - q);
  psi[idx] = 1;
  cd u[16], du[16];
  for (int g = 0; g < n_gates; ++g) {
    if (gates[g].kind == G_I) continue;
    gate_matrices(gates[g].kind, exponent_of(&gates[g], params), u, du);
    apply_gate_matrix(psi, n, &gates[g], u, 0);
  }
}

static double term_expectation(const cf* psi, int n, uint64_t x, uint64_t z, int ny) {
  const size_t dim = (size_t)1 << n;
  double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc) if (g_inner)
  for (size_t j = 0; j < dim; ++j) acc += creal(conjf(psi[j ^ x]) * pauli_phase(ny, j, z) * psi[j]);
  return acc;
}

static int team_size(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void choose_threading(int n, int U, int n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  g_inner = team_size() > 1 && (n >= 26 || (n >= 18 && 2 * U < team_size()));
}

int qo_expectation(int n, int n_gates, const qo_gate* gates, const float* params, const int8_t* bits, int U,
                   int n_ops, const int32_t* term_offsets, const float* coeffs, const uint64_t* xq,
                   const uint64_t* zq, float* out, int n_threads) {
  const size_t dim = (size_t)1 << n;
  choose_threading(n, U, n_threads);
#pragma omp parallel if (!g_inner)
  {
    cf* psi = (cf*)malloc(dim * sizeof(cf));
#pragma omp for schedule(dynamic, 1)
    for (int u = 0; u < U; ++u) {
      simulate(psi, n, n_gates, gates, params, bits + (size_t)u * n);
      for (int k = 0; k < n_ops; ++k) {
        double e = 0.0;
        for (int j = term_offsets[k]; j < term_offsets[k + 1]; ++j) {
          uint64_t x = to_index_mask(xq[j], n), z = to_index_mask(zq[j], n);
          e += coeffs[j] * term_expectation(psi, n, x, z, __builtin_popcountll(xq[j] & zq[j]));
        }
        out[(size_t)u * n_ops + k] = (float)e;
      }
    }
    free(psi);
  }
  return 0;
}

/* values + grad[p] = sum_{u,k} upstream[u,k] d out[u,k] / d params[p]  (adjoint) */
int qo_expectation_vjp(int n, int n_gates, const qo_gate* gates, const float* params, const int8_t* bits,
                       int U, int n_ops, const int32_t* term_offsets, const float* coeffs,
                       const uint64_t* xq, const uint64_t* zq, const float* upstream, float* out_vals,
                       float* grad, int n_params, int n_threads) {
  const size_t dim = (size_t)1 << n;
  double* gsum = (double*)calloc((size_t)n_params + 1, sizeof(double));
  choose_threading(n, U, n_threads);
#pragma omp parallel if (!g_inner)
  {
    cf* psi = (cf*)malloc(dim * sizeof(cf));
    cf* lam = (cf*)malloc(dim * sizeof(cf));
    double* gloc = (double*)calloc((size_t)n_params + 1, sizeof(double));
#pragma omp for schedule(dynamic, 1)
    for (int u = 0; u < U; ++u) {
      simulate(psi, n, n_gates, gates, params, bits + (size_t)u * n);
      memset(lam, 0, dim * sizeof(cf));
      for (int k = 0; k < n_ops; ++k) {
        double e = 0.0;
        const float w = upstream[(size_t)u * n_ops + k];
        for (int j = term_offsets[k]; j < term_offsets[k + 1]; ++j) {
          uint64_t x = to_index_mask(xq[j], n), z = to_index_mask(zq[j], n);
          int ny = __builtin_popcountll(xq[j] & zq[j]);
          e += coeffs[j] * term_expectation(psi, n, x, z, ny);
          const float c = w * coeffs[j];
#pragma omp parallel for schedule(static) if (g_inner)
          for (size_t i = 0; i < dim; ++i) lam[i] += c * pauli_phase(ny, i ^ x, z) * psi[i ^ x];
        }
        if (out_vals) out_vals[(size_t)u * n_ops + k] = (float)e;
      }
      cd um[16], dum[16];
      for (int g = n_gates - 1; g >= 0; --g) {
        if (gates[g].kind == G_I) continue;
        gate_matrices(gates[g].kind, exponent_of(&gates[g], params), um, dum);
        apply_gate_matrix(psi, n, &gates[g], um, 1); /* psi_{g-1} */
        if (gates[g].param_idx >= 0)
          gloc[gates[g].param_idx] += (double)gates[g].scalar * inner_du(lam, psi, n, &gates[g], dum);
        apply_gate_matrix(lam, n, &gates[g], um, 1);
      }
    }
#pragma omp critical
    for (int p = 0; p < n_params; ++p) gsum[p] += gloc[p];
    free(gloc);
    free(psi);
    free(lam);
  }
  for (int p = 0; p < n_params; ++p) grad[p] = (float)gsum[p];
  free(gsum);
  return 0;
}

/* Final states C(params)|x_u>, complex64 [U, 2^n] (the checker of qhbm_statevector at sizes numpy needs minutes for).
 * Gate records carry no cirq global_shift here: exact for the X / Z / CZ powers of the HEA (global_shift 0). */
int qo_statevector(int n, int n_gates, const qo_gate* gates, const float* params, const int8_t* bits, int U, cf* out,
                   int n_threads) {
  const size_t dim = (size_t)1 << n;
  choose_threading(n, U, n_threads);
#pragma omp parallel for schedule(dynamic, 1) if (!g_inner)
  for (int u = 0; u < U; ++u) simulate(out + (size_t)u * dim, n, n_gates, gates, params, bits + (size_t)u * n);
  return 0;
}

int qo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
