/*
 * qhbm_cpu.c -- fp32 CPU restatement (C + OpenMP) of the expectation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  It is the C twin of
 * oracle/qhbm_oracle.py and serves as a second, independently written CHECKER
 * (bench.py's `parity_check`, the GPU tests).  The TIMED CPU baseline of
 * bench.py (`cpu_baseline.kind` = "port+diag") is qhbm_cpu_diag.c, the same
 * algorithm with merged diagonal runs and an AVX2 one-qubit kernel, itself held
 * to this file.  The product path (qhbm-library_amd/) never links or calls either.
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
#include "qhbm_cpu_common.h"

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

int qo_max_threads(void) { return team_size(); }
