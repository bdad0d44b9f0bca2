// schedule.cpp -- circuit lowering and LDS-tile pass scheduling (host, no HIP).
//
// Scheduling model.  Every gate touches a few amplitude-index bits.  A pass
// keeps K "local" bits resident in LDS, so it can apply any non-diagonal gate
// whose bits are all local, and ANY diagonal gate (a diagonal gate never moves
// data: non-local bits are constants of the tile).  Gates on disjoint bits
// commute, so a pass greedily absorbs every gate whose per-bit predecessors
// have been absorbed -- for nearest-neighbour ansaetze this eats a whole
// light-cone of the circuit in one HBM round trip, not one layer.
#include "schedule.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <sstream>

#include "../../include/qhbm_engine.h"

namespace qhbm {
namespace {

inline int popc(uint32_t x) { return __builtin_popcount(x); }

bool lower(const Model& m, int n_eff, std::vector<LoweredOp>* ops, int* n_angles,
           std::string* err) {
  (void)n_eff;
  ops->clear();
  *n_angles = 0;
  for (size_t g = 0; g < m.gates.size(); ++g) {
    const Gate& G = m.gates[g];
    LoweredOp op;
    op.kind = G.kind;
    op.gate = static_cast<int>(g);
    const bool two = G.kind >= QHBM_GATE_CZPOW && G.kind < QHBM_GATE_KIND_COUNT;
    if (G.kind < 0 || G.kind >= QHBM_GATE_KIND_COUNT) {
      *err = "gate " + std::to_string(g) + ": unknown kind " + std::to_string(G.kind);
      return false;
    }
    if (G.q0 < 0 || G.q0 >= m.n || (two && (G.q1 < 0 || G.q1 >= m.n || G.q1 == G.q0))) {
      *err = "gate " + std::to_string(g) + ": qubit out of range";
      return false;
    }
    if (G.param_idx >= m.n_params) {
      *err = "gate " + std::to_string(g) + ": param_idx out of range";
      return false;
    }
    op.b0 = m.n - 1 - G.q0;
    op.b1 = two ? m.n - 1 - G.q1 : -1;
    op.bits = (1u << op.b0) | (two ? (1u << op.b1) : 0u);
    switch (G.kind) {
      case QHBM_GATE_I: op.type = LOW_SKIP; break;
      case QHBM_GATE_ZPOW:
      case QHBM_GATE_CZPOW: op.type = LOW_DIAG; op.par = false; break;
      case QHBM_GATE_ZZPOW: op.type = LOW_DIAG; op.par = true; break;
      case QHBM_GATE_XPOW:
      case QHBM_GATE_YPOW:
      case QHBM_GATE_HPOW: op.type = LOW_MAT1; break;
      default: op.type = LOW_MAT2; break;
    }
    if (op.type == LOW_DIAG) op.angle_idx = (*n_angles)++;
    if (op.type != LOW_SKIP) ops->push_back(op);
  }
  return true;
}

// Ops of `order` (indices into ops) not yet done that a pass with local set S
// can absorb, respecting per-bit order.  Returns them in execution order.
std::vector<int> absorb(const std::vector<LoweredOp>& ops, const std::vector<int>& order,
                        const std::vector<char>& done, uint32_t S, uint32_t all_bits,
                        int* n_mat) {
  std::vector<int> out;
  uint32_t blocked = 0;
  *n_mat = 0;
  for (int oi : order) {
    if (done[oi]) continue;
    const LoweredOp& op = ops[oi];
    if (op.bits & blocked) { blocked |= op.bits; }
    else if (op.type == LOW_DIAG || (op.bits & ~S) == 0) {
      out.push_back(oi);
      if (op.type != LOW_DIAG) ++*n_mat;
    } else {
      blocked |= op.bits;
    }
    if ((blocked & all_bits) == all_bits) break;
  }
  return out;
}

struct MeasGroup {
  uint32_t x;
  std::vector<int> terms;  // indices into Model::terms
};

class Builder {
 public:
  Builder(const Model& m, int K, int R, int n_eff, bool adjoint, Plan* plan)
      : m_(m), K_(K), R_(R), n_eff_(n_eff), adjoint_(adjoint), plan_(plan) {}

  Pass begin_pass(uint32_t S) {
    Pass p;
    p.K = K_;
    p.R = R_;
    for (int b = 0; b < n_eff_; ++b) {
      if (S >> b & 1) p.local_pos.push_back(b); else p.nonlocal_pos.push_back(b);
    }
    int c = 0;
    while (c < K_ && p.local_pos[c] == c) ++c;
    p.c = c;
    p.spread.resize(size_t(1) << (K_ - c));
    for (uint32_t j = 0; j < p.spread.size(); ++j) {
      uint32_t v = 0;
      for (int i = c; i < K_; ++i) if (j >> (i - c) & 1) v |= 1u << p.local_pos[i];
      p.spread[j] = v;
    }
    return p;
  }

  static uint32_t to_local(const Pass& p, uint32_t gbits) {
    uint32_t l = 0;
    for (size_t i = 0; i < p.local_pos.size(); ++i) if (gbits >> p.local_pos[i] & 1) l |= 1u << i;
    return l;
  }
  static uint32_t local_set_mask(const Pass& p) {
    uint32_t s = 0;
    for (int b : p.local_pos) s |= 1u << b;
    return s;
  }

  // Packs the absorbed ops into rounds / diagonal ops and appends the words.
  void emit_ops(Pass* p, const std::vector<LoweredOp>& ops, const std::vector<int>& absorbed) {
    std::vector<char> emitted(absorbed.size(), 0);
    size_t left = absorbed.size();
    const uint32_t S = local_set_mask(*p);
    while (left) {
      // ---- gate rounds until no gate is ready --------------------------------
      for (;;) {
        uint32_t blocked = 0, reg = 0;
        std::vector<size_t> in_round;
        for (size_t i = 0; i < absorbed.size(); ++i) {
          if (emitted[i]) continue;
          const LoweredOp& op = ops[absorbed[i]];
          if (op.bits & blocked) { blocked |= op.bits; continue; }
          if (op.type != LOW_MAT1) { blocked |= op.bits; continue; }
          const uint32_t lb = to_local(*p, op.bits);
          if (popc(reg | lb) <= R_) { reg |= lb; in_round.push_back(i); }
          else blocked |= op.bits;
        }
        if (in_round.empty()) break;
        // pad the register mask to exactly R bits, highest free local bits first
        for (int i = K_ - 1; i >= 0 && popc(reg) < R_; --i) if (!(reg >> i & 1)) reg |= 1u << i;
        p->prog.push_back(OP_ROUND | (uint32_t(in_round.size()) << 8));
        p->prog.push_back(reg);
        for (size_t i : in_round) {
          const LoweredOp& op = ops[absorbed[i]];
          emit_micro(p, op, reg);
          emitted[i] = 1;
          --left;
        }
        ++p->n_rounds;
      }
      // ---- dense two-qubit gates that are ready (applied directly on LDS) ---------
      {
        uint32_t blocked2 = 0;
        for (size_t i = 0; i < absorbed.size(); ++i) {
          if (emitted[i]) continue;
          const LoweredOp& op = ops[absorbed[i]];
          if (op.bits & blocked2) { blocked2 |= op.bits; continue; }
          if (op.type == LOW_MAT2) { emit_gate2(p, op); emitted[i] = 1; --left; }
          else blocked2 |= op.bits;
        }
      }
      // ---- one diagonal op with every ready diagonal term -----------------------
      std::vector<size_t> diag;
      uint32_t blocked = 0;
      for (size_t i = 0; i < absorbed.size(); ++i) {
        if (emitted[i]) continue;
        const LoweredOp& op = ops[absorbed[i]];
        if (op.bits & blocked) { blocked |= op.bits; continue; }
        if (op.type == LOW_DIAG) diag.push_back(i); else blocked |= op.bits;
      }
      if (!diag.empty()) {
        emit_diag(p, ops, absorbed, diag, S);
        for (size_t i : diag) { emitted[i] = 1; --left; }
      }
    }
  }

  int new_slot(Pass* p, const LoweredOp& op) {
    if (!adjoint_) return -1;
    const Gate& G = m_.gates[op.gate];
    if (G.param_idx < 0) return -1;
    const int slot = static_cast<int>(plan_->slot_gate.size());
    plan_->slot_gate.push_back(op.gate);
    plan_->slot_factor.push_back(G.scalar);
    ++p->n_slots;
    return slot;
  }

  void emit_micro(Pass* p, const LoweredOp& op, uint32_t reg) {
    const Gate& G = m_.gates[op.gate];
    auto rank = [&](int gbit) {
      uint32_t lb = to_local(*p, 1u << gbit);
      return popc(reg & (lb - 1));
    };
    CoefJob job{};
    job.op_kind = op.kind;
    job.gate = op.gate;
    job.param_idx = G.param_idx;
    job.scalar = G.scalar;
    job.offset = G.offset;
    job.out_off = plan_->n_coef_floats;
    job.dagger = adjoint_ ? 1 : 0;
    uint32_t mop;
    int nfloat;
    const uint32_t rb0 = rank(op.b0);
    if (op.kind == QHBM_GATE_XPOW) { mop = MOP_X; nfloat = 2; }
    else if (op.kind == QHBM_GATE_YPOW) { mop = MOP_Y; nfloat = 2; }
    else { mop = MOP_MAT1; nfloat = adjoint_ ? 16 : 8; }
    job.mop = mop;
    plan_->n_coef_floats += nfloat;
    plan_->jobs.push_back(job);
    const int slot = new_slot(p, op);
    p->prog.push_back(mop | (rb0 << 8) | (uint32_t(op.kind) << 16));
    p->prog.push_back(uint32_t(job.out_off));
    p->prog.push_back(uint32_t(slot));
    ++p->n_mat_ops;
  }

  void emit_gate2(Pass* p, const LoweredOp& op) {
    const Gate& G = m_.gates[op.gate];
    CoefJob job{};
    job.op_kind = op.kind;
    job.mop = MOP_MAT2;
    job.gate = op.gate;
    job.param_idx = G.param_idx;
    job.scalar = G.scalar;
    job.offset = G.offset;
    job.out_off = plan_->n_coef_floats;
    job.dagger = adjoint_ ? 1 : 0;
    plan_->n_coef_floats += adjoint_ ? 64 : 32;
    plan_->jobs.push_back(job);
    auto local_bit = [&](int gbit) { return uint32_t(__builtin_ctz(to_local(*p, 1u << gbit))); };
    const int slot = new_slot(p, op);
    p->prog.push_back(OP_GATE2 | (uint32_t(op.kind) << 8));
    p->prog.push_back(local_bit(op.b0) | (local_bit(op.b1) << 8));
    p->prog.push_back(uint32_t(job.out_off));
    p->prog.push_back(uint32_t(slot));
    ++p->n_mat_ops;
    ++p->n_rounds;
  }

  void emit_diag(Pass* p, const std::vector<LoweredOp>& ops, const std::vector<int>& absorbed,
                 const std::vector<size_t>& diag, uint32_t S) {
    const uint32_t lo_mask = (1u << std::min(kLoBits, K_)) - 1;
    std::vector<uint32_t> cls[3];  // lo, hi(+const), cross
    for (size_t i : diag) {
      const LoweredOp& op = ops[absorbed[i]];
      const uint32_t lm = to_local(*p, op.bits & S);
      const uint32_t nm = op.bits & ~S;
      int k;
      if (lm == 0 || (lm & lo_mask) == 0) k = 1;
      else if ((lm & ~lo_mask) == 0) k = 0;
      else k = 2;
      const int slot = new_slot(p, op);
      cls[k].push_back(lm | (op.par ? 0x80000000u : 0u));
      cls[k].push_back(nm);
      cls[k].push_back(uint32_t(op.angle_idx));
      cls[k].push_back(uint32_t(slot));
      ++p->n_diag_terms;
    }
    // at most kMaxCrossTerms cross terms per OP_DIAG (LDS table); overflow goes to
    // follow-up OP_DIAGs whose lo/hi tables are identically one.
    const size_t n_cross = cls[2].size() / kDiagTermWords;
    size_t done_cross = 0;
    bool first = true;
    do {
      const size_t take = std::min(n_cross - done_cross, size_t(kMaxCrossTerms));
      const uint32_t n_lo = first ? uint32_t(cls[0].size() / kDiagTermWords) : 0u;
      const uint32_t n_hi = first ? uint32_t(cls[1].size() / kDiagTermWords) : 0u;
      p->prog.push_back(OP_DIAG);
      p->prog.push_back(n_lo | n_hi << 10 | uint32_t(take) << 20);
      if (first) {
        p->prog.insert(p->prog.end(), cls[0].begin(), cls[0].end());
        p->prog.insert(p->prog.end(), cls[1].begin(), cls[1].end());
      }
      p->prog.insert(p->prog.end(), cls[2].begin() + done_cross * kDiagTermWords,
                     cls[2].begin() + (done_cross + take) * kDiagTermWords);
      done_cross += take;
      first = false;
      ++p->n_diag_ops;
    } while (done_cross < n_cross);
  }

  void emit_measure(Pass* p, const std::vector<MeasGroup>& groups, const std::vector<int>& which) {
    if (which.empty()) return;
    const uint32_t S = local_set_mask(*p);
    p->prog.push_back(OP_MEASURE | (uint32_t(which.size()) << 8));
    for (int gi : which) {
      const MeasGroup& g = groups[gi];
      p->prog.push_back(to_local(*p, g.x));
      p->prog.push_back(uint32_t(g.terms.size()));
      for (int ti : g.terms) {
        const PauliTerm& t = m_.terms[ti];
        uint32_t cb;
        std::memcpy(&cb, &t.coeff, 4);
        p->prog.push_back(to_local(*p, t.z & S));
        p->prog.push_back(t.z & ~S);
        p->prog.push_back(cb);
        p->prog.push_back(uint32_t(t.op) | (uint32_t(t.ny & 3) << 24));
        ++p->n_meas_terms;
      }
      ++p->n_meas_groups;
    }
  }

 private:
  const Model& m_;
  int K_, R_, n_eff_;
  bool adjoint_;
  Plan* plan_;
};

}  // namespace

bool build_plan(const Model& m, int tile_bits, int round_bits, bool adjoint, Plan* plan, std::string* err) {
  *plan = Plan();
  if (m.n < 1 || m.n > kMaxQubits - 1) { *err = "n_qubits must be in [1, 31]"; return false; }
  const int n_eff = std::max(m.n, kMinTileBits);
  const int k_cap = adjoint ? kMaxTileBits - 1 : kMaxTileBits;
  int K;
  if (tile_bits == 0) {
    K = n_eff <= k_cap ? n_eff : (adjoint ? 12 : 13);
  } else {
    if (tile_bits < kMinTileBits || tile_bits > k_cap) {
      *err = "tile_qubits out of range";
      return false;
    }
    K = std::min(n_eff, tile_bits);
  }
  int R = adjoint ? 4 : round_bits_for(K);
  if (!adjoint && round_bits != 0) {
    if ((round_bits != 4 && round_bits != 5) || (round_bits == 5 && K < 12)) {
      *err = "round_qubits must be 4 or 5 (5 needs tile_qubits >= 12)";
      return false;
    }
    R = round_bits;
  }
  plan->n = m.n;
  plan->n_eff = n_eff;
  plan->K = K;
  plan->R = R;
  plan->adjoint = adjoint;

  std::vector<LoweredOp> ops;
  if (!lower(m, n_eff, &ops, &plan->n_angles, err)) return false;
  // diagonal-angle jobs (shared by every pass that evaluates the term)
  for (const LoweredOp& op : ops) {
    if (op.type != LOW_DIAG) continue;
    const Gate& G = m.gates[op.gate];
    CoefJob j{};
    j.op_kind = op.kind; j.mop = 0; j.gate = op.gate; j.param_idx = G.param_idx;
    j.scalar = G.scalar; j.offset = G.offset; j.out_off = op.angle_idx;
    plan->jobs.push_back(j);
  }
  std::vector<int> order(ops.size());
  for (size_t i = 0; i < ops.size(); ++i) order[i] = adjoint ? int(ops.size() - 1 - i) : int(i);

  const uint32_t all_bits = n_eff >= 32 ? 0xFFFFFFFFu : ((1u << n_eff) - 1);
  const int c_min = std::min(K, 4);
  Builder b(m, K, R, n_eff, adjoint, plan);
  std::vector<char> done(ops.size(), 0);
  size_t n_done = 0;

  while (n_done < ops.size()) {
    // ---- candidate local sets -------------------------------------------------
    std::vector<uint32_t> cands;
    if (K >= n_eff) {
      cands.push_back(all_bits);
    } else {
      const uint32_t low = (1u << c_min) - 1;
      const int h = K - c_min;
      for (int p = c_min; p + h <= n_eff; ++p) cands.push_back(low | (((1u << h) - 1) << p));
      // demand-driven: bits of the earliest ready non-diagonal ops
      uint32_t S = low, blocked = 0;
      for (int oi : order) {
        if (done[oi]) continue;
        const LoweredOp& op = ops[oi];
        if (op.bits & blocked) { blocked |= op.bits; continue; }
        if (op.type == LOW_DIAG) continue;
        if (popc(S | op.bits) <= K) S |= op.bits; else blocked |= op.bits;
      }
      for (int bit = n_eff - 1; bit >= 0 && popc(S) < K; --bit) if (!(S >> bit & 1)) S |= 1u << bit;
      cands.push_back(S);
    }
    uint32_t best_S = 0;
    int best_mat = -1;
    size_t best_total = 0;
    std::vector<int> best_list;
    for (uint32_t S : cands) {
      int n_mat = 0;
      std::vector<int> lst = absorb(ops, order, done, S, all_bits, &n_mat);
      if (n_mat > best_mat || (n_mat == best_mat && lst.size() > best_total)) {
        best_mat = n_mat; best_total = lst.size(); best_S = S; best_list.swap(lst);
      }
    }
    if (best_list.empty()) { *err = "scheduler made no progress"; return false; }
    if (adjoint) {  // bound the gradient slots one pass owns (LDS accumulators)
      size_t keep = 0;
      int slots = 0;
      for (; keep < best_list.size(); ++keep) {
        if (m.gates[ops[best_list[keep]].gate].param_idx >= 0 && ++slots > kMaxSlotsPerPass) break;
      }
      best_list.resize(keep);
    }
    Pass p = b.begin_pass(best_S);
    p.slot_base = int(plan->slot_gate.size());
    b.emit_ops(&p, ops, best_list);
    for (int oi : best_list) { done[oi] = 1; ++n_done; }
    plan->passes.push_back(std::move(p));
  }

  if (adjoint) {
    for (Pass& p : plan->passes) {
      p.flags = PASS_ADJOINT | PASS_STORE;
      p.prog.push_back(OP_END);
    }
    if (!plan->passes.empty()) plan->passes.back().flags &= ~PASS_STORE;
    return true;
  }

  // ---- forward: measurement --------------------------------------------------
  if (plan->passes.empty()) {
    const uint32_t S = K >= n_eff ? all_bits : ((1u << K) - 1);
    plan->passes.push_back(b.begin_pass(S));
  }
  plan->passes.front().flags |= PASS_INIT_BASIS;
  plan->passes.back().completes_circuit = true;

  std::vector<MeasGroup> groups;
  for (size_t ti = 0; ti < m.terms.size(); ++ti) {
    const uint32_t x = m.terms[ti].x;
    auto it = std::find_if(groups.begin(), groups.end(), [&](const MeasGroup& g) { return g.x == x; });
    if (it == groups.end()) { groups.push_back(MeasGroup{x, {}}); it = groups.end() - 1; }
    it->terms.push_back(int(ti));
  }
  std::vector<char> gdone(groups.size(), 0);
  size_t g_left = groups.size();
  auto take = [&](Pass* p) {
    uint32_t S = 0;
    for (int bb : p->local_pos) S |= 1u << bb;
    std::vector<int> which;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      if (!gdone[gi] && (groups[gi].x & ~S) == 0) { which.push_back(int(gi)); gdone[gi] = 1; --g_left; }
    }
    b.emit_measure(p, groups, which);
  };
  take(&plan->passes.back());
  while (g_left) {
    // measurement-only pass: local set = low bits + the x-bits of as many groups as fit
    uint32_t S = 1u;  // bit 0 is always local (amplitudes move in 16-byte pairs)
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      if (gdone[gi]) continue;
      if (popc(S | groups[gi].x) <= K) S |= groups[gi].x;
    }
    for (int bit = 0; bit < n_eff && popc(S) < K; ++bit) if (!(S >> bit & 1)) S |= 1u << bit;
    Pass p = b.begin_pass(S);
    p.is_measure_only = true;
    const size_t before = g_left;
    take(&p);
    if (g_left == before) {
      *err = "a Pauli term flips more qubits than fit in one tile (" + std::to_string(K - 1) + ")";
      return false;
    }
    plan->passes.push_back(std::move(p));
  }
  for (Pass& p : plan->passes) p.prog.push_back(OP_END);
  return true;
}

std::string describe_plan(const Plan& p) {
  std::ostringstream os;
  os << (p.adjoint ? "adjoint" : "forward") << " plan: n=" << p.n << " n_eff=" << p.n_eff
     << " tile_bits=" << p.K << " round_bits=" << p.R << " passes=" << p.passes.size()
     << " coef_floats=" << p.n_coef_floats << " angles=" << p.n_angles << "\n";
  for (size_t i = 0; i < p.passes.size(); ++i) {
    const Pass& q = p.passes[i];
    os << "  pass " << i << ": c=" << q.c << " local=[";
    for (size_t k = 0; k < q.local_pos.size(); ++k) os << (k ? "," : "") << q.local_pos[k];
    os << "] mat_ops=" << q.n_mat_ops << " rounds=" << q.n_rounds << " diag_ops=" << q.n_diag_ops
       << " diag_terms=" << q.n_diag_terms << " meas_groups=" << q.n_meas_groups
       << " meas_terms=" << q.n_meas_terms << " slots=" << q.n_slots
       << (q.is_measure_only ? " [measure-only]" : "") << " words=" << q.prog.size() << "\n";
  }
  return os.str();
}

}  // namespace qhbm
