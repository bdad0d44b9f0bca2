// plan_fuzz.cpp -- randomized structural check of the HOST scheduler (csrc/schedule.cpp, the planning half of
// csrc/engine.cpp), built with -fsanitize=address,undefined by tests/sanitize/Makefile and run by
// tests/test_sanitize_cpu.py.  No device is touched: engines are created with device = -1 (planning only).
//
// Test infrastructure, not product.  SURVEY.md section 5 asks for a sanitizer run of the CPU build; the index
// arithmetic of the scheduler decides every launch and had only example-based tests.
//
// For every random circuit (all twelve gate kinds, constant and parametrised exponents, shared parameters, random
// observables) and every option set (tile size, relabeling, wave-bit mapping, FULL threshold, wide last pass):
//   * every lowered micro-op is scheduled exactly once: the multiset of coefficient jobs (gate, micro-op, multiplier)
//     does not depend on the tiling, forward and adjoint plans agree on it;
//   * each pass is well formed: local + non-local bits partition the index, programs parse to OP_END, records and
//     thread tables lie inside their buffers, a round's thread table is a bijection onto the free local bits,
//     boundary-phase predicates name a free local bit or a tile bit;
//   * adjoint plans: <= kMaxSlotsPerPass gradient slots per pass, every slot of a pass written exactly once, every
//     trainable gate owns at least one slot, frozen and constant gates own none;
//   * through the C ABI: pass counts > 0, flop / traffic models and the micro-op census finite and >= 0, and a
//     plan rebuilt after gradient-mask changes equals the plan of a fresh engine given the final mask.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/qhbm_engine.h"
#include "../../qhbm-library_amd/csrc/schedule.h"

using namespace qhbm;

static int g_failures = 0;
static std::string g_context;
#define CHECK(cond, ...)                                                        \
  do {                                                                          \
    if (!(cond)) {                                                              \
      if (g_failures < 20) {                                                    \
        std::fprintf(stderr, "FAIL %s:%d [%s] %s : ", __FILE__, __LINE__, g_context.c_str(), #cond); \
        std::fprintf(stderr, __VA_ARGS__);                                      \
        std::fprintf(stderr, "\n");                                             \
      }                                                                         \
      ++g_failures;                                                             \
    }                                                                           \
  } while (0)

struct Case {
  Model m;
  std::vector<qhbm_gate> abi_gates;
  std::vector<int32_t> term_offsets;
  std::vector<float> coeffs;
  std::vector<uint64_t> xs, zs;
};

static Case random_case(std::mt19937_64& rng, int n) {
  Case c;
  auto U = [&](int lo, int hi) { return int(std::uniform_int_distribution<int>(lo, hi)(rng)); };
  auto F = [&](float lo, float hi) { return std::uniform_real_distribution<float>(lo, hi)(rng); };
  const int style = U(0, 3);
  const int n_params = U(1, 24);
  std::vector<qhbm_gate> gates;
  auto add = [&](int kind, int q0, int q1, bool constant) {
    qhbm_gate g{};
    g.kind = kind;
    g.q0 = q0;
    g.q1 = q1;
    g.param_idx = constant ? -1 : U(0, n_params - 1);
    g.scalar = constant ? 0.f : (U(0, 3) ? 1.f : F(-2.f, 2.f));
    g.offset = constant ? (U(0, 2) ? 1.f : F(-1.f, 1.f)) : (U(0, 4) ? 0.f : F(-1.f, 1.f));
    g.global_shift = U(0, 5) ? 0.f : -0.5f;
    gates.push_back(g);
  };
  if (style == 0) {  // hardware-efficient layers (config 3's shape)
    const int layers = U(1, n >= 20 ? 3 : 6);
    for (int l = 0; l < layers; ++l) {
      for (int q = 0; q < n; ++q) { add(QHBM_GATE_XPOW, q, -1, false); add(QHBM_GATE_ZPOW, q, -1, false); }
      for (int q = 0; q + 1 < n; q += 2) add(QHBM_GATE_CZPOW, q, q + 1, false);
      for (int q = 1; q + 1 < n; q += 2) add(QHBM_GATE_CZPOW, q, q + 1, false);
    }
  } else {  // anything goes
    const int n_gates = U(1, n >= 20 ? 40 : 80);
    for (int i = 0; i < n_gates; ++i) {
      const int kind = U(style == 1 ? 1 : 0, QHBM_GATE_KIND_COUNT - 1);
      const bool two = kind >= QHBM_GATE_CZPOW;
      const int q0 = U(0, n - 1);
      int q1 = -1;
      if (two) {
        if (n < 2) continue;
        do { q1 = (style == 2 && U(0, 1)) ? std::min(n - 1, q0 + 1) : U(0, n - 1); } while (q1 == q0 && (q1 = (q0 + 1) % n, false));
        if (q1 == q0) q1 = (q0 + 1) % n;
      }
      add(kind, q0, q1, U(0, 4) == 0);
    }
  }
  c.abi_gates = gates;
  c.m.n = n;
  c.m.n_params = n_params;
  for (const qhbm_gate& g : gates) c.m.gates.push_back(Gate{g.kind, g.q0, g.q1, g.param_idx, g.scalar, g.offset, g.global_shift});
  // observables
  const int n_ops = U(1, 3);
  c.m.n_ops = n_ops;
  c.term_offsets.push_back(0);
  const uint64_t all = n >= 32 ? ~0ull : ((1ull << n) - 1ull);
  for (int op = 0; op < n_ops; ++op) {
    const int shape = U(0, 3);
    int n_terms = shape == 3 ? U(32, 70) : U(1, 9);
    for (int t = 0; t < n_terms; ++t) {
      uint64_t x = 0, z = 0;
      if (shape == 0) {  // chain-like: one or two flips, neighbouring Z
        const int q = U(0, n - 1);
        x = U(0, 1) ? (1ull << q) : 0ull;
        z = (1ull << q) | (U(0, 1) ? (1ull << ((q + 1) % n)) : 0ull);
        if (U(0, 1)) z &= ~x;
      } else if (shape == 3) {  // many diagonal strings (the Walsh-Hadamard measurement)
        z = rng() & all;
      } else {
        x = rng() & rng() & all;
        z = rng() & all;
        if (U(0, 2) == 0) x = rng() & all;
      }
      c.coeffs.push_back(F(-1.f, 1.f));
      c.xs.push_back(x);
      c.zs.push_back(z);
    }
    c.term_offsets.push_back(int32_t(c.coeffs.size()));
  }
  return c;
}

typedef std::tuple<int, int, int, int> JobKey;  // gate, micro-op, gate kind, multiplier in 1/4096
static std::map<JobKey, int> job_multiset(const Plan& p) {
  std::map<JobKey, int> s;
  for (const CoefJob& j : p.jobs) ++s[JobKey(j.gate, j.mop, j.op_kind, int(std::lround(double(j.mult) * 4096.0)))];
  return s;
}

static void check_plan(const Model& m, const Plan& plan) {
  CHECK(plan.n_eff >= m.n && plan.n_eff >= kMinTileBits && plan.n_eff <= kMaxQubits, "n_eff %d n %d", plan.n_eff, m.n);
  CHECK(plan.K >= kMinTileBits && plan.K <= kMaxTileBits && plan.K <= plan.n_eff, "K %d", plan.K);
  CHECK(plan.R == kRoundBits, "R %d", plan.R);
  const RecordLayout L(plan.R, plan.adjoint);
  CHECK(plan.coef_init.size() <= size_t(plan.n_coef_floats) + 4096, "coef_init %zu floats %d", plan.coef_init.size(), plan.n_coef_floats);
  for (uint32_t off : plan.record_offsets) CHECK(size_t(off) + size_t(L.words()) <= plan.coef_init.size(), "record at %u", off);
  if (plan.adjoint) CHECK(plan.slot_gate.size() == plan.slot_factor.size(), "slot tables");
  int slots_total = 0;
  for (size_t pi = 0; pi < plan.passes.size(); ++pi) {
    const Pass& p = plan.passes[pi];
    const int K = p.K;
    CHECK(K >= kMinTileBits && K <= kMaxTileBits && p.R == plan.R, "pass %zu K %d", pi, K);
    CHECK(int(p.local_pos.size()) == K && int(p.local_pos.size() + p.nonlocal_pos.size()) == plan.n_eff, "pass %zu bit sets", pi);
    uint32_t seen = 0, seen_phys = 0;
    for (int b : p.local_pos) { CHECK(b >= 0 && b < plan.n_eff && !(seen >> b & 1), "local bit %d", b); seen |= 1u << b; }
    for (int b : p.nonlocal_pos) { CHECK(b >= 0 && b < plan.n_eff && !(seen >> b & 1), "nonlocal bit %d", b); seen |= 1u << b; }
    for (int b : p.local_phys) { CHECK(b >= 0 && b < plan.n_eff && !(seen_phys >> b & 1), "local phys %d", b); seen_phys |= 1u << b; }
    for (int b : p.nonlocal_phys) { CHECK(b >= 0 && b < plan.n_eff && !(seen_phys >> b & 1), "nonlocal phys %d", b); seen_phys |= 1u << b; }
    CHECK(std::is_sorted(p.local_phys.begin(), p.local_phys.end()) && std::is_sorted(p.nonlocal_phys.begin(), p.nonlocal_phys.end()), "phys order");
    CHECK(p.c >= 0 && p.c <= K && p.spread.size() == (size_t(1) << (K - p.c)), "pass %zu c %d", pi, p.c);
    CHECK(p.round_tl.size() % (size_t(1) << (K - p.R)) == 0, "tl tables");
    if (plan.adjoint) {
      CHECK(p.n_slots >= 0 && p.n_slots <= kMaxSlotsPerPass, "pass %zu: %d slots", pi, p.n_slots);
      CHECK(p.slot_base == slots_total, "pass %zu slot_base %d expected %d", pi, p.slot_base, slots_total);
      slots_total += p.n_slots;
    }
    std::vector<int> slot_hits(size_t(std::max(p.n_slots, 0)), 0);
    bool ended = false;
    size_t pc = 0, rounds = 0;
    while (pc < p.prog.size()) {
      const uint32_t w0 = p.prog[pc], opc = w0 & 0xffu;
      if (opc == OP_END) { ended = true; break; }
      if (opc == OP_ROUND) {
        CHECK(pc + kRoundWords <= p.prog.size(), "round words");
        if (pc + kRoundWords > p.prog.size()) break;
        const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8, reg = p.prog[pc + 1], first = p.prog[pc + 2], tl = p.prog[pc + 3], dead = p.prog[pc + 4];
        CHECK(__builtin_popcount(reg) == p.R && (reg >> K) == 0, "regmask %x", reg);
        CHECK(n_inst >= 1 && size_t(first) + size_t(n_inst) * size_t(L.words()) <= plan.coef_init.size(), "records %u x %u", first, n_inst);
        const size_t nt = size_t(1) << (K - p.R);
        CHECK(size_t(tl) + nt <= p.round_tl.size(), "tl table %u", tl);
        CHECK((dead & reg) == 0 && (dead >> K) == 0 && (plan.adjoint || dead == 0), "dead mask %x", dead);
        if (size_t(tl) + nt <= p.round_tl.size()) {
          std::set<uint32_t> distinct;
          for (size_t t = 0; t < nt; ++t) {
            const uint32_t v = p.round_tl[tl + t];
            CHECK((v & reg) == 0 && (v >> K) == 0, "TL[%zu] = %x reg %x", t, v, reg);
            distinct.insert(v);
          }
          CHECK(distinct.size() == nt, "TL not a bijection: %zu of %zu", distinct.size(), nt);
        }
        if (size_t(first) + size_t(n_inst) * size_t(L.words()) <= plan.coef_init.size())
          for (uint32_t i = 0; i < n_inst; ++i) {
            const uint32_t* rec = &plan.coef_init[first + size_t(i) * size_t(L.words())];
            const uint32_t h1 = rec[1];
            for (int k = 0; k < 8; ++k) {
              if (!(h1 >> k & 1u)) continue;
              const uint32_t pred = rec[L.pred(k)], pos = pred & 0xffu;
              if (pred >> 8) CHECK(pos >= uint32_t(K) && pos < uint32_t(plan.n_eff), "tile predicate %u", pos);
              else CHECK(pos < uint32_t(K) && !(reg >> pos & 1u), "thread predicate %u reg %x", pos, reg);
            }
            if (plan.adjoint)
              for (int w = 0; w < 32; ++w) {
                const uint32_t s = rec[L.slot0() + w];
                if (s == 0xffffffffu) continue;
                CHECK(s < uint32_t(p.n_slots), "slot %u of %d", s, p.n_slots);
                if (s < uint32_t(p.n_slots)) ++slot_hits[s];
              }
          }
        ++rounds;
        pc += kRoundWords;
      } else if (opc == OP_GATE2) {
        CHECK(pc + kGate2Words <= p.prog.size(), "gate2 words");
        if (pc + kGate2Words > p.prog.size()) break;
        const uint32_t pw = p.prog[pc + 1], p0 = pw & 0xffu, p1 = (pw >> 8) & 0xffu;
        CHECK(p0 < uint32_t(K) && p1 < uint32_t(K) && p0 != p1, "gate2 bits %u %u", p0, p1);
        CHECK(size_t(p.prog[pc + 2]) + (plan.adjoint ? 64 : 32) <= size_t(plan.n_coef_floats), "gate2 coefficients");
        const uint32_t s = p.prog[pc + 3];
        if (plan.adjoint && s != 0xffffffffu) {
          CHECK(s < uint32_t(p.n_slots), "gate2 slot %u", s);
          if (s < uint32_t(p.n_slots)) ++slot_hits[s];
        }
        pc += kGate2Words;
      } else if (opc == OP_MEASURE_WHT) {
        const size_t n_terms = w0 >> 8;
        CHECK(!plan.adjoint && pc + kWhtHeaderWords + n_terms * kMeasTermWords <= p.prog.size(), "wht group");
        if (pc + kWhtHeaderWords + n_terms * kMeasTermWords > p.prog.size()) break;
        uint32_t prev = 0;
        for (int cidx = 0; cidx < 16; ++cidx) { const uint32_t e = p.prog[pc + 1 + cidx]; CHECK(e >= prev && e <= n_terms, "class_end"); prev = e; }
        for (size_t t = 0; t < n_terms; ++t) CHECK(p.prog[pc + kWhtHeaderWords + t * kMeasTermWords + 3] < uint32_t(m.n_ops), "wht op index");
        // a term sits in the class the KERNEL of this pass reads it from: its local z mask cut at the pass's K - 4
        // (the wide last pass of a forward plan was cut at the plan's K: wrong values at 19 qubits, round 5)
        for (size_t t = 0; t < n_terms; ++t) {
          const uint32_t zl = p.prog[pc + kWhtHeaderWords + t * kMeasTermWords];
          uint32_t cls = 0;
          while (cls < 15u && t >= p.prog[pc + 1 + cls]) ++cls;
          CHECK((zl >> K) == 0 && (zl >> (K - 4)) == cls, "wht term %zu: z mask %x in class %u of a tile of 2^%d", t, zl, cls, K);
        }
        pc += size_t(kWhtHeaderWords) + n_terms * kMeasTermWords;
      } else if (opc == OP_MEASURE) {
        CHECK(!plan.adjoint, "measurement in an adjoint plan");
        const uint32_t n_groups = w0 >> 8;
        ++pc;
        bool bad = false;
        for (uint32_t g = 0; g < n_groups && !bad; ++g) {
          if (pc + 2 > p.prog.size()) { bad = true; break; }
          const uint32_t xl = p.prog[pc], n_terms = p.prog[pc + 1];
          CHECK((xl >> K) == 0, "measure x mask %x", xl);
          pc += 2;
          if (pc + size_t(n_terms) * kMeasTermWords > p.prog.size()) { bad = true; break; }
          for (uint32_t t = 0; t < n_terms; ++t) CHECK((p.prog[pc + t * kMeasTermWords + 3] & 0xffffffu) < uint32_t(m.n_ops), "op index");
          pc += size_t(n_terms) * kMeasTermWords;
        }
        CHECK(!bad, "measure group runs past the program");
        if (bad) break;
      } else {
        CHECK(false, "unknown opcode %u at %zu", opc, pc);
        break;
      }
    }
    CHECK(ended, "pass %zu: program does not reach OP_END", pi);
    if (plan.adjoint)
      for (int s = 0; s < p.n_slots; ++s) CHECK(slot_hits[size_t(s)] == 1, "pass %zu slot %d written %d times", pi, s, slot_hits[size_t(s)]);
  }
  if (plan.adjoint) {
    CHECK(slots_total == int(plan.slot_gate.size()), "slots %d table %zu", slots_total, plan.slot_gate.size());
    std::vector<char> has(m.gates.size(), 0);
    for (int g : plan.slot_gate) {
      CHECK(g >= 0 && size_t(g) < m.gates.size(), "slot gate %d", g);
      if (g >= 0 && size_t(g) < m.gates.size()) {
        has[size_t(g)] = 1;
        CHECK(m.gates[size_t(g)].param_idx >= 0 && !m.frozen(m.gates[size_t(g)].param_idx), "slot on a constant / frozen gate %d", g);
      }
    }
    for (const float f : plan.slot_factor) CHECK(std::isfinite(f), "slot factor");
  }
}

// the gates that must own a slot: trainable, not frozen, not lowered away (identity, zero scalar keeps its slot)
static void check_trainable_covered(const Model& m, const Plan& adj, const std::map<JobKey, int>& jobs) {
  std::set<int> gates_with_jobs;
  for (const auto& kv : jobs) gates_with_jobs.insert(std::get<0>(kv.first));
  std::set<int> with_slot(adj.slot_gate.begin(), adj.slot_gate.end());
  if (adj.dense_tail || !m.param_frozen.empty()) return;  // (a sweep that stops early owns no slots for the leading gates)
  for (size_t g = 0; g < m.gates.size(); ++g) {
    const Gate& G = m.gates[g];
    if (G.param_idx < 0 || G.kind == QHBM_GATE_I) continue;
    if (!gates_with_jobs.count(int(g))) continue;
    CHECK(with_slot.count(int(g)), "trainable gate %zu (kind %d) has no gradient slot", g, G.kind);
  }
}

static int fuzz_one(std::mt19937_64& rng, int n, int index) {
  auto U = [&](int lo, int hi) { return int(std::uniform_int_distribution<int>(lo, hi)(rng)); };
  Case c = random_case(rng, n);
  // observables in the scheduler's form (engine.cpp set_observables: qubit q <-> index bit n - 1 - q is done there; the
  // masks here are already in index space)
  for (int op = 0; op < c.m.n_ops; ++op)
    for (int j = c.term_offsets[size_t(op)]; j < c.term_offsets[size_t(op) + 1]; ++j) {
      PauliTerm t;
      t.coeff = c.coeffs[size_t(j)];
      t.x = uint32_t(c.xs[size_t(j)]);
      t.z = uint32_t(c.zs[size_t(j)]);
      t.ny = __builtin_popcount(t.x & t.z);
      t.op = op;
      c.m.terms.push_back(t);
    }
  char ctx[128];
  std::map<JobKey, int> ref_jobs;
  bool have_ref = false;
  const int n_eff = std::max(n, kMinTileBits);
  const int option_sets = 3;
  for (int os = 0; os < option_sets; ++os) {
    const int tile = os == 0 ? 0 : std::min(n_eff, U(kMinTileBits, kMaxTileBits - 1));
    const bool relabel = U(0, 1), wave_bits = U(0, 3) != 0;
    const int full_threshold = (const int[]){0, 60, 60, 100000}[U(0, 3)];
    const int wide = U(-1, 1);
    for (int adjoint = 0; adjoint < 2; ++adjoint) {
      std::snprintf(ctx, sizeof ctx, "case %d n=%d gates=%zu tile=%d relabel=%d wave=%d full=%d wide=%d adj=%d", index, n, c.m.gates.size(),
                    tile, int(relabel), int(wave_bits), full_threshold, wide, adjoint);
      g_context = ctx;
      Plan plan;
      std::string err;
      const int tb = adjoint ? std::min(tile, 13) : tile;  // (adjoint kernels exist for 2^10 .. 2^13)
      const bool ok = build_plan(c.m, tb, kRoundBits, adjoint != 0, &plan, &err, full_threshold, 0, wave_bits, adjoint && relabel, wide, nullptr);
      CHECK(ok, "build_plan: %s", err.c_str());
      if (!ok) continue;
      check_plan(c.m, plan);
      const std::map<JobKey, int> jobs = job_multiset(plan);
      if (!have_ref) { ref_jobs = jobs; have_ref = true; }
      else CHECK(jobs == ref_jobs, "the scheduled micro-ops differ from the first plan's (%zu vs %zu kinds)", jobs.size(), ref_jobs.size());
      if (adjoint) check_trainable_covered(c.m, plan, jobs);
      (void)describe_plan(plan);
    }
  }
  // ---- through the C ABI (planning-only engine) ----
  g_context = std::string("case ") + std::to_string(index) + " abi";
  qhbm_engine* h = nullptr;
  CHECK(qhbm_create(-1, &h) == 0 && h, "create");
  if (!h) return 0;
  // qubit-space masks for the ABI: bit q of the mask <-> qubit q; the engine maps to index bits itself
  int rc = qhbm_set_circuit(h, n, int(c.abi_gates.size()), c.abi_gates.data(), c.m.n_params);
  CHECK(rc == 0, "set_circuit: %s", qhbm_last_error(h));
  rc = qhbm_set_observables(h, c.m.n_ops, c.term_offsets.data(), c.coeffs.data(), c.xs.data(), c.zs.data());
  CHECK(rc == 0, "set_observables: %s", qhbm_last_error(h));
  const int tq = U(0, 2) == 0 ? std::min(n_eff, U(10, 13)) : 0, atq = U(0, 2) == 0 ? std::min(n_eff, U(10, 13)) : 0;
  if (tq) CHECK(qhbm_set_option(h, "tile_qubits", tq) == 0, "tile_qubits %d: %s", tq, qhbm_last_error(h));
  if (atq) CHECK(qhbm_set_option(h, "adjoint_tile_qubits", atq) == 0, "adjoint_tile_qubits %d: %s", atq, qhbm_last_error(h));
  int fp = 0, bp = 0;
  rc = qhbm_num_passes(h, &fp, &bp);
  CHECK(rc == 0 && fp >= 1 && bp >= 0, "num_passes rc %d (%s) %d %d", rc, qhbm_last_error(h), fp, bp);
  double a = -1, b = -1, d = -1;
  rc = qhbm_flop_model(h, 8, 1, &a, &b, &d);
  CHECK(rc == 0 && std::isfinite(a) && std::isfinite(b) && std::isfinite(d) && a >= 0 && b >= 0 && d >= 0, "flop model %g %g %g", a, b, d);
  rc = qhbm_traffic_model(h, 8, 1, &a, &b, &d);
  CHECK(rc == 0 && std::isfinite(a) && std::isfinite(b) && std::isfinite(d) && a >= 0 && b > 0 && d >= 0, "traffic model %g %g %g", a, b, d);
  std::vector<double> census(size_t(64) * QHBM_CENSUS_COLUMNS);
  for (int adjoint = 0; adjoint < 2; ++adjoint) {
    int np = 0;
    rc = qhbm_op_census(h, adjoint, 64, census.data(), &np);
    CHECK(rc == 0 && np == (adjoint ? bp : fp), "census rc %d passes %d", rc, np);
    for (double v : census) CHECK(std::isfinite(v) && v >= 0, "census entry %g", v);
  }
  std::vector<char> text(1 << 16);
  CHECK(qhbm_describe_schedule(h, text.data(), text.size()) == 0 && text[0], "describe");
  // gradient masks: change twice, then compare with a fresh engine given the final mask
  std::vector<uint8_t> m1(size_t(c.m.n_params)), m2(size_t(c.m.n_params));
  for (auto& v : m1) v = uint8_t(U(0, 1));
  for (auto& v : m2) v = uint8_t(U(0, 3) != 0);
  CHECK(qhbm_set_gradient_mask(h, m1.data(), c.m.n_params) == 0, "mask 1");
  CHECK(qhbm_num_passes(h, &fp, &bp) == 0, "plan under mask 1: %s", qhbm_last_error(h));
  CHECK(qhbm_set_gradient_mask(h, m2.data(), c.m.n_params) == 0, "mask 2");
  CHECK(qhbm_set_gradient_mask(h, m1.data(), c.m.n_params) == 0, "mask 1 again");
  CHECK(qhbm_num_passes(h, &fp, &bp) == 0, "plan under mask 1 again");
  CHECK(qhbm_set_gradient_mask(h, m2.data(), c.m.n_params) == 0, "mask 2 again");
  std::vector<char> after(1 << 16), fresh(1 << 16);
  CHECK(qhbm_describe_schedule(h, after.data(), after.size()) == 0, "describe after masks: %s", qhbm_last_error(h));
  {
    qhbm_engine* f = nullptr;
    CHECK(qhbm_create(-1, &f) == 0 && f, "create fresh");
    if (f) {
      CHECK(qhbm_set_circuit(f, n, int(c.abi_gates.size()), c.abi_gates.data(), c.m.n_params) == 0, "fresh circuit");
      CHECK(qhbm_set_observables(f, c.m.n_ops, c.term_offsets.data(), c.coeffs.data(), c.xs.data(), c.zs.data()) == 0, "fresh observables");
      if (tq) CHECK(qhbm_set_option(f, "tile_qubits", tq) == 0, "fresh tile_qubits");
      if (atq) CHECK(qhbm_set_option(f, "adjoint_tile_qubits", atq) == 0, "fresh adjoint_tile_qubits");
      CHECK(qhbm_set_gradient_mask(f, m2.data(), c.m.n_params) == 0, "fresh mask");
      CHECK(qhbm_describe_schedule(f, fresh.data(), fresh.size()) == 0, "fresh describe: %s", qhbm_last_error(f));
      CHECK(std::string(after.data()) == std::string(fresh.data()), "the plan after gradient-mask changes differs from a fresh engine's");
      qhbm_destroy(f);
    }
  }
  qhbm_destroy(h);
  return 0;
}

// The checker checks: corrupted copies of a valid plan must each be reported.
static int self_test() {
  std::mt19937_64 rng(7);
  Case c = random_case(rng, 14);
  c.m.gates.clear();
  for (int l = 0; l < 3; ++l) {
    for (int q = 0; q < 14; ++q) { c.m.gates.push_back(Gate{QHBM_GATE_XPOW, q, -1, q, 1.f, 0.f, 0.f}); c.m.gates.push_back(Gate{QHBM_GATE_ZPOW, q, -1, q, 1.f, 0.f, 0.f}); }
    for (int q = 0; q + 1 < 14; ++q) c.m.gates.push_back(Gate{QHBM_GATE_CZPOW, q, q + 1, q, 1.f, 0.f, 0.f});
  }
  c.m.n_params = 14;
  c.m.terms.clear();
  c.m.terms.push_back(PauliTerm{1.f, 0u, 3u, 0, 0});
  c.m.n_ops = 1;
  Plan plan;
  std::string err;
  if (!build_plan(c.m, 10, kRoundBits, true, &plan, &err)) { std::fprintf(stderr, "self test: %s\n", err.c_str()); return 1; }
  g_context = "self test";
  int missed = 0;
  auto expect_failure = [&](const char* what, Plan q) {
    const int before = g_failures;
    g_failures = 1000;  // silence the messages of the expected failures
    check_plan(c.m, q);
    const bool caught = g_failures > 1000;
    g_failures = before;
    if (!caught) { std::fprintf(stderr, "self test: corruption not detected: %s\n", what); ++missed; }
  };
  check_plan(c.m, plan);
  if (g_failures) { std::fprintf(stderr, "self test: the valid plan fails\n"); return 1; }
  { Plan q = plan; q.passes[0].n_slots += 1; expect_failure("slot count", q); }
  { Plan q = plan; q.passes[0].round_tl[1] = q.passes[0].round_tl[0]; expect_failure("thread table not a bijection", q); }
  { Plan q = plan; q.passes[0].prog.back() = OP_ROUND; expect_failure("program without OP_END", q); }
  { Plan q = plan; q.passes[0].prog[1] |= 1u << 15; expect_failure("register mask outside the tile", q); }
  { Plan q = plan; std::swap(q.passes[0].local_pos[0], q.passes[0].nonlocal_pos[0]); q.passes[0].local_pos[1] = q.passes[0].local_pos[0]; expect_failure("bit sets", q); }
  { Plan q = plan; const RecordLayout L(q.R, true); q.coef_init[q.passes[0].prog[2] + L.slot0()] = 0; q.coef_init[q.passes[0].prog[2] + L.slot0() + 1] = 0; expect_failure("slot written twice", q); }
  {
    Plan q = plan;
    q.jobs.pop_back();
    if (job_multiset(q) == job_multiset(plan)) { std::fprintf(stderr, "self test: a dropped micro-op is not detected\n"); ++missed; }
  }
  return missed;
}

int main(int argc, char** argv) {
  if (self_test()) { std::printf("plan_fuzz: self test FAILED\n"); return 2; }
  const int cases = argc > 1 ? std::atoi(argv[1]) : 2000;
  const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 20261003ull;
  const int n_max = argc > 3 ? std::atoi(argv[3]) : 28;
  std::mt19937_64 rng(seed);
  int by_n[40] = {0};
  for (int i = 0; i < cases; ++i) {
    // n = 3 .. n_max; the large sizes are rarer (their plans take longer under the sanitizers)
    int n = 3 + int(rng() % uint64_t(n_max - 2));
    if (n > 20 && (rng() & 3)) n = 3 + int(rng() % 18);
    ++by_n[n];
    fuzz_one(rng, n, i);
    if (g_failures > 200) break;
  }
  std::printf("plan_fuzz: %d cases (seed %llu), qubit counts", cases, (unsigned long long)seed);
  for (int n = 3; n <= n_max; ++n) std::printf(" %d:%d", n, by_n[n]);
  std::printf("\nplan_fuzz: %d failures\n", g_failures);
  return g_failures ? 1 : 0;
}
