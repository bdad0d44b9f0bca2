/*
 * qhbm_cpu_diag.c -- the TIMED CPU baseline BASELINE.md section 3 describes ("port+diag"): the same fp32 statevector
 * algorithm as qhbm_cpu.c (the gate-by-gate checker), with what a CPU simulator of this class does beside it:
 *
 *   * runs of consecutive DIAGONAL gates (Z**t, CZ**t, ZZ**t: an HEA layer holds 2n-1 of them) are merged into ONE phase
 *     sweep  psi[j] *= T_hi[j >> h] * T_lo[v(j)][j & (2^h - 1)]  (two half-index tables; the few pairs that straddle the
 *     halves select one of <= 16 variants of the low table), and in the adjoint sweep into ONE pass over (psi, lambda)
 *     that un-applies the phases and returns every gradient of the run from two histograms of Im(conj(lambda) psi);
 *   * one-qubit dense gates (X / Y / H powers) run through an AVX2 kernel (four complex floats per vector), and their
 *     adjoint step -- un-apply on psi, 2 Re<lambda| dU |psi>, un-apply on lambda -- is ONE fused pass instead of three;
 *   * Pauli terms that share an X mask are measured, and added to lambda, in one pass per mask;
 *   * threads across states below 26 qubits, inside a state from 26 on (and when a call holds few states): TFQ's policy.
 *
 * No gate fusion beyond the per-layer diagonal merging (BASELINE.md section 3 says so): two-qubit non-diagonal gates go
 * gate by gate through the checker's own sweeps.
 *
 * TEST INFRASTRUCTURE / reported baseline, NOT THE PRODUCT.  Held to qhbm_cpu.c in tests/test_oracle_c.py (values 1e-6,
 * gradients 1e-5 relative).  What it restates: tfq.layers.Expectation forward + adjoint as called at
 * /root/reference/qhbmlib/inference/qnn.py:134-138 (qnn.py:90-99 for the differentiator).
 */
#include "qhbm_cpu_common.h"

#include <immintrin.h>

int qo_expectation_vjp(int n, int n_gates, const qo_gate* gates, const float* params, const int8_t* bits, int U, int n_ops,
                       const int32_t* term_offsets, const float* coeffs, const uint64_t* xq, const uint64_t* zq,
                       const float* upstream, float* out_vals, float* grad, int n_params, int n_threads);

/* ---- four complex floats per vector ------------------------------------------------------------------------------- */
static inline __m256 cmul_s(__m256 a, __m256 mr, __m256 mi) { /* a * (mr + i mi), broadcast scalar */
  return _mm256_fmaddsub_ps(a, mr, _mm256_mul_ps(_mm256_permute_ps(a, 0xB1), mi));
}
static inline __m256 cmul_v(__m256 a, __m256 b) { /* element-wise complex product */
  return _mm256_fmaddsub_ps(a, _mm256_moveldup_ps(b), _mm256_mul_ps(_mm256_permute_ps(a, 0xB1), _mm256_movehdup_ps(b)));
}
static inline double hsum8(__m256 v) {
  __m128 s = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
  s = _mm_add_ps(s, _mm_movehl_ps(s, s));
  s = _mm_add_ss(s, _mm_shuffle_ps(s, s, 1));
  return (double)_mm_cvtss_f32(s);
}

typedef struct { __m256 r[4], i[4]; } mat2v;
static mat2v splat2(const cf* m) {
  mat2v o;
  for (int k = 0; k < 4; ++k) { o.r[k] = _mm256_set1_ps(crealf(m[k])); o.i[k] = _mm256_set1_ps(cimagf(m[k])); }
  return o;
}
/* the pair vectors (a: bit clear, b: bit set) of eight consecutive amplitudes for target bits 0 and 1 */
static inline void split_pairs(int bit, __m256 v0, __m256 v1, __m256* a, __m256* b) {
  if (bit == 1) { *a = _mm256_permute2f128_ps(v0, v1, 0x20); *b = _mm256_permute2f128_ps(v0, v1, 0x31); }
  else {
    *a = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(v0), _mm256_castps_pd(v1)));
    *b = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(v0), _mm256_castps_pd(v1)));
  }
}
static inline void join_pairs(int bit, __m256 a, __m256 b, __m256* v0, __m256* v1) {
  if (bit == 1) { *v0 = _mm256_permute2f128_ps(a, b, 0x20); *v1 = _mm256_permute2f128_ps(a, b, 0x31); }
  else {
    *v0 = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(a), _mm256_castps_pd(b)));
    *v1 = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(a), _mm256_castps_pd(b)));
  }
}
#define MATVEC(M, a, b, na, nb)                                                          \
  do {                                                                                   \
    na = _mm256_add_ps(cmul_s(a, (M).r[0], (M).i[0]), cmul_s(b, (M).r[1], (M).i[1]));     \
    nb = _mm256_add_ps(cmul_s(a, (M).r[2], (M).i[2]), cmul_s(b, (M).r[3], (M).i[3]));     \
  } while (0)

/* psi <- (m on `bit`) psi,  n >= 3 */
static void apply1_avx(cf* psi, int n, int bit, const cf* m) {
  float* p = (float*)psi;
  const size_t dim = (size_t)1 << n, half = dim >> 1, st = (size_t)1 << bit;
  const mat2v M = splat2(m);
  if (bit >= 2) {
#pragma omp parallel for schedule(static) if (g_inner)
    for (size_t i = 0; i < half; i += 4) {
      const size_t k = ((i >> bit) << (bit + 1)) | (i & (st - 1));
      __m256 a = _mm256_loadu_ps(p + 2 * k), b = _mm256_loadu_ps(p + 2 * (k + st)), na, nb;
      MATVEC(M, a, b, na, nb);
      _mm256_storeu_ps(p + 2 * k, na);
      _mm256_storeu_ps(p + 2 * (k + st), nb);
    }
  } else {
#pragma omp parallel for schedule(static) if (g_inner)
    for (size_t k = 0; k < dim; k += 8) {
      __m256 v0 = _mm256_loadu_ps(p + 2 * k), v1 = _mm256_loadu_ps(p + 2 * k + 8), a, b, na, nb;
      split_pairs(bit, v0, v1, &a, &b);
      MATVEC(M, a, b, na, nb);
      join_pairs(bit, na, nb, &v0, &v1);
      _mm256_storeu_ps(p + 2 * k, v0);
      _mm256_storeu_ps(p + 2 * k + 8, v1);
    }
  }
}

/* One fused adjoint step of a one-qubit gate: psi <- U^dag psi, returns 2 Re <lam| dU |psi> (lam before its own un-apply,
 * psi after), lam <- U^dag lam.  Partial sums leave the float lanes every 64 vectors (double beyond). */
static double adjoint1_avx(cf* psi, cf* lam, int n, int bit, const cf* udag, const cf* du, int want_grad) {
  float* p = (float*)psi;
  float* l = (float*)lam;
  const size_t dim = (size_t)1 << n, half = dim >> 1, st = (size_t)1 << bit;
  const mat2v D = splat2(udag), G = splat2(du);
  double total = 0.0;
  const size_t n_vec = half / 4, blk = 64;
#pragma omp parallel for schedule(static) reduction(+ : total) if (g_inner)
  for (size_t v0i = 0; v0i < n_vec; v0i += blk) {
    __m256 acc = _mm256_setzero_ps();
    const size_t v1i = v0i + blk < n_vec ? v0i + blk : n_vec;
    for (size_t vi = v0i; vi < v1i; ++vi) {
      __m256 a, b, la, lb, pa, pb, ga, gb, nla, nlb;
      size_t ka, kb;   /* float offsets of the two loads */
      __m256 w0, w1, x0, x1;
      if (bit >= 2) {
        const size_t i = vi * 4, k = ((i >> bit) << (bit + 1)) | (i & (st - 1));
        ka = 2 * k; kb = 2 * (k + st);
        a = _mm256_loadu_ps(p + ka); b = _mm256_loadu_ps(p + kb);
        la = _mm256_loadu_ps(l + ka); lb = _mm256_loadu_ps(l + kb);
      } else {
        ka = 16 * vi; kb = ka + 8;
        w0 = _mm256_loadu_ps(p + ka); w1 = _mm256_loadu_ps(p + kb);
        x0 = _mm256_loadu_ps(l + ka); x1 = _mm256_loadu_ps(l + kb);
        split_pairs(bit, w0, w1, &a, &b);
        split_pairs(bit, x0, x1, &la, &lb);
      }
      MATVEC(D, a, b, pa, pb);
      if (want_grad) {
        MATVEC(G, pa, pb, ga, gb);
        acc = _mm256_fmadd_ps(la, ga, acc);   /* Re(conj(l) g) = l.re g.re + l.im g.im, lane by lane */
        acc = _mm256_fmadd_ps(lb, gb, acc);
      }
      MATVEC(D, la, lb, nla, nlb);
      if (bit >= 2) {
        _mm256_storeu_ps(p + ka, pa); _mm256_storeu_ps(p + kb, pb);
        _mm256_storeu_ps(l + ka, nla); _mm256_storeu_ps(l + kb, nlb);
      } else {
        join_pairs(bit, pa, pb, &w0, &w1);
        join_pairs(bit, nla, nlb, &x0, &x1);
        _mm256_storeu_ps(p + ka, w0); _mm256_storeu_ps(p + kb, w1);
        _mm256_storeu_ps(l + ka, x0); _mm256_storeu_ps(l + kb, x1);
      }
    }
    total += hsum8(acc);
  }
  return 2.0 * total;
}

/* ---- merged runs of diagonal gates --------------------------------------------------------------------------------- */
enum { D_Z = 0, D_CZ = 1, D_ZZ = 2 };
typedef struct { int type, a, b, gate; } diag_item;   /* index bits a (, b); `gate` = position in the gate list */
#define MAX_CROSS_BITS 4
typedef struct {
  int first, count;            /* items [first, first + count) */
  int n_xbits, xbits[MAX_CROSS_BITS];   /* high-half index bits that pair with a low-half bit */
  cf* t_hi;                    /* [2^(n-h)] */
  cf* t_lo;                    /* [2^n_xbits][2^h] */
} diag_run;

static int in_set(int condtype, int ba, int bb) { return condtype == D_Z ? ba : condtype == D_CZ ? (ba & bb) : (ba ^ bb); }

static int variant_of(const diag_run* r, size_t jh, int h) {
  int v = 0;
  for (int k = 0; k < r->n_xbits; ++k) v |= (int)((jh >> (r->xbits[k] - h)) & 1u) << k;
  return v;
}

static void build_tables(diag_run* r, const diag_item* items, const double* t_of_item, int n, int h) {
  const size_t n_hi = (size_t)1 << (n - h), n_lo = (size_t)1 << h;
  const int n_var = 1 << r->n_xbits;
  r->t_hi = (cf*)malloc(n_hi * sizeof(cf));
  r->t_lo = (cf*)malloc((size_t)n_var * n_lo * sizeof(cf));
  for (size_t jh = 0; jh < n_hi; ++jh) {
    double th = 0.0;
    for (int k = r->first; k < r->first + r->count; ++k) {
      const diag_item* it = &items[k];
      const int a_hi = it->a >= h, b_hi = it->type == D_Z ? a_hi : it->b >= h;
      if (!(a_hi && b_hi)) continue;
      const int ba = (int)((jh >> (it->a - h)) & 1u), bb = it->type == D_Z ? 0 : (int)((jh >> (it->b - h)) & 1u);
      if (in_set(it->type, ba, bb)) th += t_of_item[k];
    }
    r->t_hi[jh] = (cf)cexp(I * M_PI * th);
  }
  for (int v = 0; v < n_var; ++v)
    for (size_t jl = 0; jl < n_lo; ++jl) {
      double th = 0.0;
      for (int k = r->first; k < r->first + r->count; ++k) {
        const diag_item* it = &items[k];
        const int a_hi = it->a >= h, b_hi = it->type == D_Z ? a_hi : it->b >= h;
        if (a_hi && b_hi) continue;
        int ba, bb = 0;
        if (a_hi) { int x = 0; while (r->xbits[x] != it->a) ++x; ba = (v >> x) & 1; } else ba = (int)((jl >> it->a) & 1u);
        if (it->type != D_Z) {
          if (b_hi) { int x = 0; while (r->xbits[x] != it->b) ++x; bb = (v >> x) & 1; } else bb = (int)((jl >> it->b) & 1u);
        }
        if (in_set(it->type, ba, bb)) th += t_of_item[k];
      }
      r->t_lo[(size_t)v * n_lo + jl] = (cf)cexp(I * M_PI * th);
    }
}

/* psi <- D psi (conj = 0) or D^dag psi (conj = 1) */
static void diag_apply(cf* psi, const diag_run* r, int n, int h, int conj_) {
  const size_t n_hi = (size_t)1 << (n - h), n_lo = (size_t)1 << h;
  const __m256 sgn = conj_ ? _mm256_set_ps(-0.f, 0.f, -0.f, 0.f, -0.f, 0.f, -0.f, 0.f) : _mm256_setzero_ps();
#pragma omp parallel for schedule(static) if (g_inner)
  for (size_t jh = 0; jh < n_hi; ++jh) {
    float* row = (float*)(psi + (jh << h));
    const float* tab = (const float*)(r->t_lo + (size_t)variant_of(r, jh, h) * n_lo);
    const cf ph = conj_ ? conjf(r->t_hi[jh]) : r->t_hi[jh];
    const __m256 pr = _mm256_set1_ps(crealf(ph)), pi = _mm256_set1_ps(cimagf(ph));
    for (size_t jl = 0; jl < n_lo; jl += 4) {
      const __m256 t = _mm256_xor_ps(_mm256_loadu_ps(tab + 2 * jl), sgn);
      _mm256_storeu_ps(row + 2 * jl, cmul_v(cmul_s(_mm256_loadu_ps(row + 2 * jl), pr, pi), t));
    }
  }
}

/* The adjoint step of a run: w_j = Im(conj(lam_j) psi_j) into the histograms w_hi[2^(n-h)] and w_lo[variant][2^h]
 * (the caller turns them into the gradient of every gate of the run), then psi <- D^dag psi, lam <- D^dag lam. */
static void diag_adjoint(cf* psi, cf* lam, const diag_run* r, int n, int h, double* w_hi, float* w_lo) {
  const size_t n_hi = (size_t)1 << (n - h), n_lo = (size_t)1 << h;
  const __m256 sgn = _mm256_set_ps(-0.f, 0.f, -0.f, 0.f, -0.f, 0.f, -0.f, 0.f);
  const int n_var = 1 << r->n_xbits;
  /* (the low histogram is shared by the rows: inside a state every thread of the team fills one of its own) */
#pragma omp parallel if (g_inner)
  {
    float* mine = w_lo;
#ifdef _OPENMP
    if (omp_get_num_threads() > 1) mine = (float*)calloc((size_t)n_var * n_lo, sizeof(float));
#endif
#pragma omp for schedule(static)
    for (size_t jh = 0; jh < n_hi; ++jh) {
      float* prow = (float*)(psi + (jh << h));
      float* lrow = (float*)(lam + (jh << h));
      const int v = variant_of(r, jh, h);
      const float* tab = (const float*)(r->t_lo + (size_t)v * n_lo);
      float* wl = mine + (size_t)v * n_lo;
      const cf ph = conjf(r->t_hi[jh]);
      const __m256 pr = _mm256_set1_ps(crealf(ph)), pi = _mm256_set1_ps(cimagf(ph));
      __m128 rowacc = _mm_setzero_ps();
      for (size_t jl = 0; jl < n_lo; jl += 4) {
        const __m256 ps = _mm256_loadu_ps(prow + 2 * jl), la = _mm256_loadu_ps(lrow + 2 * jl);
        /* [lr pi, li pr] per amplitude; w = lr pi - li pr */
        const __m256 prod = _mm256_mul_ps(la, _mm256_permute_ps(ps, 0xB1));
        const __m256 hs = _mm256_hsub_ps(prod, prod);                       /* [w0 w1 w0 w1 | w2 w3 w2 w3] */
        const __m128 w = _mm_shuffle_ps(_mm256_castps256_ps128(hs), _mm256_extractf128_ps(hs, 1), 0x44);
        _mm_storeu_ps(wl + jl, _mm_add_ps(_mm_loadu_ps(wl + jl), w));
        rowacc = _mm_add_ps(rowacc, w);
        const __m256 t = _mm256_xor_ps(_mm256_loadu_ps(tab + 2 * jl), sgn);
        _mm256_storeu_ps(prow + 2 * jl, cmul_v(cmul_s(ps, pr, pi), t));
        _mm256_storeu_ps(lrow + 2 * jl, cmul_v(cmul_s(la, pr, pi), t));
      }
      rowacc = _mm_add_ps(rowacc, _mm_movehl_ps(rowacc, rowacc));
      rowacc = _mm_add_ss(rowacc, _mm_shuffle_ps(rowacc, rowacc, 1));
      w_hi[jh] += (double)_mm_cvtss_f32(rowacc);
    }
    if (mine != w_lo) {
#pragma omp critical
      for (size_t k = 0; k < (size_t)n_var * n_lo; ++k) w_lo[k] += mine[k];
      free(mine);
    }
  }
}

/* sum of w over the index set of one item, from the two histograms */
static double item_sum(const diag_run* r, const diag_item* it, int n, int h, const double* w_hi, const float* w_lo) {
  const size_t n_hi = (size_t)1 << (n - h), n_lo = (size_t)1 << h;
  const int a_hi = it->a >= h, b_hi = it->type == D_Z ? a_hi : it->b >= h;
  double s = 0.0;
  if (a_hi && b_hi) {
    for (size_t jh = 0; jh < n_hi; ++jh) {
      const int ba = (int)((jh >> (it->a - h)) & 1u), bb = it->type == D_Z ? 0 : (int)((jh >> (it->b - h)) & 1u);
      if (in_set(it->type, ba, bb)) s += w_hi[jh];
    }
    return s;
  }
  const int n_var = 1 << r->n_xbits;
  for (int v = 0; v < n_var; ++v)
    for (size_t jl = 0; jl < n_lo; ++jl) {
      int ba, bb = 0;
      if (a_hi) { int x = 0; while (r->xbits[x] != it->a) ++x; ba = (v >> x) & 1; } else ba = (int)((jl >> it->a) & 1u);
      if (it->type != D_Z) {
        if (b_hi) { int x = 0; while (r->xbits[x] != it->b) ++x; bb = (v >> x) & 1; } else bb = (int)((jl >> it->b) & 1u);
      }
      if (in_set(it->type, ba, bb)) s += (double)w_lo[(size_t)v * n_lo + jl];
    }
  return s;
}

/* ---- the program: dense gates and merged diagonal runs, in circuit order ----------------------------------------------- */
enum { OP_DENSE1 = 0, OP_GENERIC = 1, OP_DIAG = 2 };
typedef struct { int type, gate, run; } prog_op;

static int diag_type(int kind) { return kind == G_Z ? D_Z : kind == G_CZ ? D_CZ : kind == G_ZZ ? D_ZZ : -1; }

typedef struct {
  int n_ops, n_runs, n_items;
  prog_op* ops;
  diag_run* runs;
  diag_item* items;
  double* item_t;     /* exponent of every item */
  cf *m, *mdag, *du;  /* per gate: 2x2 forward matrix, its dagger, d/dt (dense one-qubit gates) */
} program;

static void build_program(program* P, int n, int h, int n_gates, const qo_gate* gates, const float* params) {
  P->ops = (prog_op*)malloc(sizeof(prog_op) * (size_t)(n_gates + 1));
  P->runs = (diag_run*)calloc((size_t)n_gates + 1, sizeof(diag_run));
  P->items = (diag_item*)malloc(sizeof(diag_item) * (size_t)(n_gates + 1));
  P->item_t = (double*)malloc(sizeof(double) * (size_t)(n_gates + 1));
  P->m = (cf*)calloc((size_t)(n_gates + 1) * 4, sizeof(cf));
  P->mdag = (cf*)calloc((size_t)(n_gates + 1) * 4, sizeof(cf));
  P->du = (cf*)calloc((size_t)(n_gates + 1) * 4, sizeof(cf));
  P->n_ops = P->n_runs = P->n_items = 0;
  int open = -1;   /* the run still accepting items */
  for (int g = 0; g < n_gates; ++g) {
    const int kind = gates[g].kind, dt = diag_type(kind);
    if (kind == G_I) continue;
    if (dt >= 0) {
      diag_item it = {dt, n - 1 - gates[g].q0, dt == D_Z ? -1 : n - 1 - gates[g].q1, g};
      /* high bits this item would add to the run's straddling set */
      int add[2], n_add = 0;
      if (dt != D_Z && ((it.a >= h) != (it.b >= h))) add[n_add++] = it.a >= h ? it.a : it.b;
      for (;;) {
        if (open < 0) {
          open = P->n_runs++;
          P->runs[open].first = P->n_items;
          P->runs[open].count = 0;
          P->runs[open].n_xbits = 0;
          P->ops[P->n_ops++] = (prog_op){OP_DIAG, -1, open};
        }
        diag_run* r = &P->runs[open];
        int fresh = 0;
        for (int k = 0; k < n_add; ++k) {
          int seen = 0;
          for (int x = 0; x < r->n_xbits; ++x) seen |= r->xbits[x] == add[k];
          fresh += !seen;
        }
        if (r->n_xbits + fresh > MAX_CROSS_BITS) { open = -1; continue; }   /* this run is full: start another */
        for (int k = 0; k < n_add; ++k) {
          int seen = 0;
          for (int x = 0; x < r->n_xbits; ++x) seen |= r->xbits[x] == add[k];
          if (!seen) r->xbits[r->n_xbits++] = add[k];
        }
        break;
      }
      P->item_t[P->n_items] = exponent_of(&gates[g], params);
      P->items[P->n_items++] = it;
      P->runs[open].count++;
      continue;
    }
    open = -1;
    if (!two_qubit(kind)) {
      cd u[16], du[16];
      gate_matrices(kind, exponent_of(&gates[g], params), u, du);
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          P->m[4 * g + 2 * i + j] = (cf)u[2 * i + j];
          P->mdag[4 * g + 2 * i + j] = (cf)conj(u[2 * j + i]);
          P->du[4 * g + 2 * i + j] = (cf)du[2 * i + j];
        }
      P->ops[P->n_ops++] = (prog_op){OP_DENSE1, g, -1};
    } else {
      P->ops[P->n_ops++] = (prog_op){OP_GENERIC, g, -1};
    }
  }
  for (int r = 0; r < P->n_runs; ++r) build_tables(&P->runs[r], P->items, P->item_t, n, h);
}

static void free_program(program* P) {
  for (int r = 0; r < P->n_runs; ++r) { free(P->runs[r].t_hi); free(P->runs[r].t_lo); }
  free(P->ops); free(P->runs); free(P->items); free(P->item_t); free(P->m); free(P->mdag); free(P->du);
}

/* ---- Pauli terms grouped by X mask ------------------------------------------------------------------------------------- */
typedef struct { uint64_t x, z; int ny, op; float c; } term_rec;
static int by_x(const void* a, const void* b) {
  const term_rec *p = (const term_rec*)a, *q = (const term_rec*)b;
  return p->x < q->x ? -1 : p->x > q->x ? 1 : 0;
}

/* values[op] += <psi|O_op|psi>;  lam += sum_op up[op] O_op psi   (lam may be NULL) */
static void observables(const cf* psi, cf* lam, int n, const term_rec* terms, int n_terms, int n_ops, const float* up,
                        double* values) {
  const size_t dim = (size_t)1 << n;
  static const cf ipow[4] = {1, I, -1, -I};
  for (int g0 = 0; g0 < n_terms;) {
    int g1 = g0;
    while (g1 < n_terms && terms[g1].x == terms[g0].x) ++g1;
    const uint64_t x = terms[g0].x;
#pragma omp parallel for schedule(static) reduction(+ : values[:n_ops]) if (g_inner)
    for (size_t j = 0; j < dim; ++j) {
      const size_t src = j ^ x;
      const cf ps = psi[src], t = conjf(psi[j]) * ps;
      cf wsum = 0;
      for (int k = g0; k < g1; ++k) {
        cf f = ipow[terms[k].ny & 3];
        if (__builtin_popcountll(src & terms[k].z) & 1) f = -f;
        values[terms[k].op] += (double)(terms[k].c * crealf(t * f));
        if (lam) wsum += (up[terms[k].op] * terms[k].c) * f;
      }
      if (lam) lam[j] += wsum * ps;
    }
    g0 = g1;
  }
}

/* ---- entry point: values [U, n_ops] + grad [P] (grad may be NULL: forward only), same contract as qo_expectation_vjp ---- */
int qo_expectation_vjp_diag(int n, int n_gates, const qo_gate* gates, const float* params, const int8_t* bits, int U,
                            int n_ops, const int32_t* term_offsets, const float* coeffs, const uint64_t* xq,
                            const uint64_t* zq, const float* upstream, float* out_vals, float* grad, int n_params,
                            int n_threads) {
  if (n < 6) {   /* (vectors of four amplitudes per half-index row need three bits per half) */
    if (grad) return qo_expectation_vjp(n, n_gates, gates, params, bits, U, n_ops, term_offsets, coeffs, xq, zq, upstream,
                                        out_vals, grad, n_params, n_threads);
    float* zero_up = (float*)calloc((size_t)U * (size_t)(n_ops > 0 ? n_ops : 1), sizeof(float));
    float* g = (float*)calloc((size_t)n_params + 1, sizeof(float));
    int rc = qo_expectation_vjp(n, n_gates, gates, params, bits, U, n_ops, term_offsets, coeffs, xq, zq, zero_up, out_vals, g,
                                n_params, n_threads);
    free(zero_up); free(g);
    return rc;
  }
  const size_t dim = (size_t)1 << n;
  const int h = n / 2;
  choose_threading(n, U, n_threads);
  program P;
  build_program(&P, n, h, n_gates, gates, params);
  const int n_terms = term_offsets[n_ops];
  term_rec* terms = (term_rec*)malloc(sizeof(term_rec) * (size_t)(n_terms + 1));
  for (int k = 0; k < n_ops; ++k)
    for (int j = term_offsets[k]; j < term_offsets[k + 1]; ++j)
      terms[j] = (term_rec){to_index_mask(xq[j], n), to_index_mask(zq[j], n), __builtin_popcountll(xq[j] & zq[j]), k, coeffs[j]};
  qsort(terms, (size_t)n_terms, sizeof(term_rec), by_x);
  double* gsum = (double*)calloc((size_t)n_params + 1, sizeof(double));
#pragma omp parallel if (!g_inner)
  {
    cf* psi = (cf*)malloc(dim * sizeof(cf));
    cf* lam = grad ? (cf*)malloc(dim * sizeof(cf)) : NULL;
    double* gloc = (double*)calloc((size_t)n_params + 1, sizeof(double));
    double* vals = (double*)malloc(sizeof(double) * (size_t)(n_ops + 1));
    double* w_hi = (double*)malloc(sizeof(double) << (n - h));
    float* w_lo = (float*)malloc((sizeof(float) << h) << MAX_CROSS_BITS);
#pragma omp for schedule(dynamic, 1)
    for (int u = 0; u < U; ++u) {
      memset(psi, 0, dim * sizeof(cf));
      size_t idx = 0;
      for (int q = 0; q < n; ++q) if (bits[(size_t)u * n + q]) idx |= (size_t)1 << (n - 1 - q);
      psi[idx] = 1;
      for (int o = 0; o < P.n_ops; ++o) {
        const prog_op* op = &P.ops[o];
        if (op->type == OP_DENSE1) apply1_avx(psi, n, n - 1 - gates[op->gate].q0, P.m + 4 * op->gate);
        else if (op->type == OP_DIAG) diag_apply(psi, &P.runs[op->run], n, h, 0);
        else {
          cd um[16], dum[16];
          gate_matrices(gates[op->gate].kind, exponent_of(&gates[op->gate], params), um, dum);
          apply_gate_matrix(psi, n, &gates[op->gate], um, 0);
        }
      }
      for (int k = 0; k < n_ops; ++k) vals[k] = 0.0;
      if (lam) memset(lam, 0, dim * sizeof(cf));
      observables(psi, lam, n, terms, n_terms, n_ops, upstream ? upstream + (size_t)u * n_ops : NULL, vals);
      if (out_vals) for (int k = 0; k < n_ops; ++k) out_vals[(size_t)u * n_ops + k] = (float)vals[k];
      if (!grad) continue;
      for (int o = P.n_ops - 1; o >= 0; --o) {
        const prog_op* op = &P.ops[o];
        if (op->type == OP_DENSE1) {
          const qo_gate* G = &gates[op->gate];
          const double d = adjoint1_avx(psi, lam, n, n - 1 - G->q0, P.mdag + 4 * op->gate, P.du + 4 * op->gate, G->param_idx >= 0);
          if (G->param_idx >= 0) gloc[G->param_idx] += (double)G->scalar * d;
        } else if (op->type == OP_DIAG) {
          const diag_run* r = &P.runs[op->run];
          memset(w_hi, 0, sizeof(double) << (n - h));
          memset(w_lo, 0, (sizeof(float) << h) << r->n_xbits);
          diag_adjoint(psi, lam, r, n, h, w_hi, w_lo);
          for (int k = r->first; k < r->first + r->count; ++k) {
            const qo_gate* G = &gates[P.items[k].gate];
            if (G->param_idx < 0) continue;
            /* d/dt: 2 Re <lam| i pi P_S |psi> = -2 pi sum_{j in S} Im(conj(lam_j) psi_j) */
            gloc[G->param_idx] += (double)G->scalar * (-2.0 * M_PI) * item_sum(r, &P.items[k], n, h, w_hi, w_lo);
          }
        } else {
          const qo_gate* G = &gates[op->gate];
          cd um[16], dum[16];
          gate_matrices(G->kind, exponent_of(G, params), um, dum);
          apply_gate_matrix(psi, n, G, um, 1);
          if (G->param_idx >= 0) gloc[G->param_idx] += (double)G->scalar * inner_du(lam, psi, n, G, dum);
          apply_gate_matrix(lam, n, G, um, 1);
        }
      }
    }
#pragma omp critical
    for (int p = 0; p < n_params; ++p) gsum[p] += gloc[p];
    free(gloc); free(vals); free(w_hi); free(w_lo); free(psi); free(lam);
  }
  if (grad) for (int p = 0; p < n_params; ++p) grad[p] = (float)gsum[p];
  free(gsum); free(terms);
  free_program(&P);
  return 0;
}
