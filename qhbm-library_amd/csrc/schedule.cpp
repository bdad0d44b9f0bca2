// schedule.cpp -- circuit lowering and LDS-tile pass scheduling (host, no HIP).
//
// Scheduling model.  Every gate touches a few amplitude-index bits.
//   * A PASS keeps K "local" bits resident in LDS.  It can apply any
//     non-diagonal gate whose bits are all local, and any diagonal term that
//     has at least one local bit (non-local bits are constants of the tile).
//     Gates on disjoint bits commute, so a pass greedily absorbs every gate
//     whose per-bit predecessors have been absorbed -- for nearest-neighbour
//     ansaetze this eats a whole light-cone of the circuit in one HBM round
//     trip, not one layer.
//   * Inside a pass, a ROUND keeps R of the K local bits in registers (2^R
//     amplitudes per thread) and again absorbs a light-cone: every one-qubit
//     gate on a register bit and every diagonal term with a register bit whose
//     predecessors are done.  The micro-ops of a round are packed into fixed
//     template INSTANCES (program.h) so the kernel body is straight-line code
//     with predicated in-place slots.
#include "schedule.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "../../include/qhbm_engine.h"

namespace qhbm {
namespace {

inline int popc(uint32_t x) { return __builtin_popcount(x); }

// A gate with a constant, odd-integer exponent: CNOT**(+-1) = CNOT and H**(+-1) = H exactly (eigenvalues +-1).
bool odd_constant(const Gate& G) {
  if (G.param_idx >= 0) return false;
  const float t = G.offset;
  return t == std::nearbyint(t) && (long(std::nearbyint(t)) & 1L);
}

bool lower(const Model& m, std::vector<LoweredOp>* ops, std::string* err, double* const_phase = nullptr,
           std::vector<std::pair<int, float>>* phases = nullptr) {
  ops->clear();
  if (const_phase) *const_phase = 0.0;
  if (phases) phases->clear();
  const bool fuse = !std::getenv("QHBM_NO_SANDWICH_FUSION");
  // A constant Hadamard (H**k, k odd) is  e^{-i pi/4} Z^(1/2) X^(1/2) Z^(1/2): three LEAN ops instead of a dense 2 x 2
  // that would turn its pass over to the general kernels; a constant CNOT is H_t CZ H_t.  The X**t kernels leave out
  // cirq's e^{i pi t / 2} (restored for qhbm_statevector from the jobs), so a lowered H owes the exported state
  // e^{-i pi/4} more: `const_phase`, in units of pi.  Exponents: the ops carry the GATE's exponent k times `mult`.
  const bool lean_clifford = !std::getenv("QHBM_NO_LEAN_CLIFFORD");
  auto push_hadamard = [&](int bit, int gate, float k, bool fixed_exponents = false) {
    LoweredOp z;
    z.kind = QHBM_GATE_ZPOW;
    z.type = LOW_DIAG;
    z.gate = gate;
    z.b0 = bit;
    z.bits = 1u << bit;
    z.mult = 0.5f / k;
    if (fixed_exponents) {  // (inside the decomposition of a PARAMETRISED gate: the exponents 1/2 are constants of their own)
      z.mult = 1.f;
      z.fixed = true;
      z.fixed_t = 0.5f;
    }
    LoweredOp x = z;
    x.kind = QHBM_GATE_XPOW;
    x.type = LOW_MAT1;
    ops->push_back(z);
    ops->push_back(x);
    ops->push_back(z);
    if (const_phase) *const_phase -= 0.25;
  };
  // (P x P)**(mult t) for P = X (which = 0), Y (1), Z (2) on the two bits of `op`: the ZZ power's three phases carry the
  // gate's exponent; XX = (H x H) ZZ (H x H), YY = (S x S) XX (S x S)^dagger with constants of their own.
  auto push_pair_power = [&](const LoweredOp& op, int which, float mult) {
    auto push_s = [&](int bit, float t) {
      LoweredOp sgt;
      sgt.kind = QHBM_GATE_ZPOW;
      sgt.type = LOW_DIAG;
      sgt.gate = op.gate;
      sgt.b0 = bit;
      sgt.bits = 1u << bit;
      sgt.fixed = true;
      sgt.fixed_t = t;
      ops->push_back(sgt);
    };
    if (which == 1) { push_s(op.b0, -0.5f); push_s(op.b1, -0.5f); }
    if (which <= 1) { push_hadamard(op.b0, op.gate, 1.f, true); push_hadamard(op.b1, op.gate, 1.f, true); }
    LoweredOp a = op, b = op, c = op;  // e^{i pi t [b0 xor b1]} = e^{i pi t b0} e^{i pi t b1} e^{-2 i pi t b0 b1}
    a.kind = b.kind = c.kind = QHBM_GATE_ZZPOW;
    a.type = b.type = c.type = LOW_DIAG;
    a.bits = 1u << op.b0; a.b1 = -1;
    b.bits = 1u << op.b1; b.b0 = op.b1; b.b1 = -1;
    a.mult = b.mult = mult;
    c.mult = -2.f * mult;
    ops->push_back(a);
    ops->push_back(b);
    ops->push_back(c);
    if (which <= 1) { push_hadamard(op.b0, op.gate, 1.f, true); push_hadamard(op.b1, op.gate, 1.f, true); }
    if (which == 1) { push_s(op.b0, 0.5f); push_s(op.b1, 0.5f); }
  };
  for (size_t g = 0; g < m.gates.size(); ++g) {
    // tfq.util.exponential writes exp(-i theta Z_a Z_b / 2) as CNOT(a, b) rz_b(theta) CNOT(a, b) and
    // exp(-i theta X_q / 2) as H_q rz_q(theta) H_q (circuit.py:268-272: the QAIA ansatz is made of these).  Taken
    // literally that is two dense two-qubit / one-qubit gates around every rotation; but CNOT Z_b^t CNOT = (Z_a Z_b)^t
    // and H Z^t H = X^t EXACTLY (global phases included: ZPow(t, shift) becomes ZZPow / XPow(t, shift)), which are
    // lean ops of the fast kernels.  The fused op keeps the ROTATION's gate index: its parameter, its gradient slot
    // and its parameter-shift programs are the rotation's.
    if (fuse && g + 2 < m.gates.size()) {
      const Gate &A = m.gates[g], &Z = m.gates[g + 1], &B = m.gates[g + 2];
      const bool in_range = Z.q0 >= 0 && Z.q0 < m.n && A.q0 >= 0 && A.q0 < m.n && Z.param_idx < m.n_params;
      if (in_range && Z.kind == QHBM_GATE_ZPOW && A.kind == B.kind && A.q0 == B.q0 && A.q1 == B.q1 && odd_constant(A) &&
          odd_constant(B)) {
        LoweredOp op;
        op.gate = static_cast<int>(g + 1);
        if (A.kind == QHBM_GATE_CNOTPOW && A.q1 == Z.q0 && A.q1 >= 0 && A.q1 < m.n && A.q1 != A.q0) {
          op.kind = QHBM_GATE_ZZPOW;
          op.b0 = m.n - 1 - A.q0;
          op.b1 = m.n - 1 - A.q1;
          op.bits = (1u << op.b0) | (1u << op.b1);
          LoweredOp a = op, b = op, c = op;  // e^{i pi t [b0 xor b1]} = e^{i pi t b0} e^{i pi t b1} e^{-2 i pi t b0 b1}
          a.type = b.type = c.type = LOW_DIAG;
          a.bits = 1u << op.b0; a.b1 = -1;
          b.bits = 1u << op.b1; b.b0 = op.b1; b.b1 = -1;
          c.mult = -2.f;
          ops->push_back(a);
          ops->push_back(b);
          ops->push_back(c);
          g += 2;
          continue;
        }
        if (A.kind == QHBM_GATE_HPOW && A.q0 == Z.q0) {
          op.kind = QHBM_GATE_XPOW;
          op.type = LOW_MAT1;
          op.b0 = m.n - 1 - Z.q0;
          op.bits = 1u << op.b0;
          ops->push_back(op);
          g += 2;
          continue;
        }
      }
    }
    const Gate& G = m.gates[g];
    if (G.kind < 0 || G.kind >= QHBM_GATE_KIND_COUNT) {
      *err = "gate " + std::to_string(g) + ": unknown kind " + std::to_string(G.kind);
      return false;
    }
    const bool two = G.kind >= QHBM_GATE_CZPOW;
    if (G.q0 < 0 || G.q0 >= m.n || (two && (G.q1 < 0 || G.q1 >= m.n || G.q1 == G.q0))) {
      *err = "gate " + std::to_string(g) + ": qubit out of range";
      return false;
    }
    if (G.param_idx >= m.n_params) {
      *err = "gate " + std::to_string(g) + ": param_idx out of range";
      return false;
    }
    LoweredOp op;
    op.kind = G.kind;
    op.gate = static_cast<int>(g);
    op.b0 = m.n - 1 - G.q0;
    op.b1 = two ? m.n - 1 - G.q1 : -1;
    op.bits = (1u << op.b0) | (two ? (1u << op.b1) : 0u);
    switch (G.kind) {
      case QHBM_GATE_I: break;
      case QHBM_GATE_ZPOW:   // diag(1, e^{i pi t})
      case QHBM_GATE_CZPOW:  // diag(1, 1, 1, e^{i pi t})
        op.type = LOW_DIAG;
        ops->push_back(op);
        break;
      case QHBM_GATE_ZZPOW: {
        // e^{i pi t [b0 xor b1]} = e^{i pi t b0} e^{i pi t b1} e^{-2 i pi t b0 b1}
        LoweredOp a = op, b = op, c = op;
        a.type = b.type = c.type = LOW_DIAG;
        a.bits = 1u << op.b0; a.b1 = -1;
        b.bits = 1u << op.b1; b.b0 = op.b1; b.b1 = -1;
        c.mult = -2.f;
        ops->push_back(a);
        ops->push_back(b);
        ops->push_back(c);
        break;
      }
      case QHBM_GATE_HPOW:
        if (lean_clifford && odd_constant(G)) { push_hadamard(op.b0, op.gate, G.offset); break; }
        if (lean_clifford) {
          // H = Ry(pi/4) Z Ry(-pi/4): H**t = Y^(1/4) Z**t Y^(-1/4) (the phases of the two Y powers cancel), and
          // Y^a = S X^a S^dagger -- the S, S^dagger next to Z**t cancel in the folding pass below
          auto push_fixed = [&](int type, int kind, float t) {
            LoweredOp f = op;
            f.kind = kind;
            f.type = type;
            f.fixed = true;
            f.fixed_t = t;
            ops->push_back(f);
          };
          push_fixed(LOW_DIAG, QHBM_GATE_ZPOW, -0.5f);
          push_fixed(LOW_MAT1, QHBM_GATE_XPOW, -0.25f);
          push_fixed(LOW_DIAG, QHBM_GATE_ZPOW, 0.5f);
          LoweredOp z = op;
          z.kind = QHBM_GATE_ZPOW;
          z.type = LOW_DIAG;
          ops->push_back(z);
          push_fixed(LOW_DIAG, QHBM_GATE_ZPOW, -0.5f);
          push_fixed(LOW_MAT1, QHBM_GATE_XPOW, 0.25f);
          push_fixed(LOW_DIAG, QHBM_GATE_ZPOW, 0.5f);
          break;
        }
        op.type = LOW_MAT1;
        ops->push_back(op);
        break;
      case QHBM_GATE_YPOW:
        if (lean_clifford) {  // Y**t = S X**t S^dagger (S = Z^(1/2)): exact, cirq's phase included; the X op is the gate's
          LoweredOp s = op;
          s.kind = QHBM_GATE_ZPOW;
          s.type = LOW_DIAG;
          s.fixed = true;
          s.fixed_t = -0.5f;
          ops->push_back(s);
          op.kind = QHBM_GATE_XPOW;
          op.type = LOW_MAT1;
          ops->push_back(op);
          s.fixed_t = 0.5f;
          ops->push_back(s);
          break;
        }
        op.type = LOW_MAT1;
        ops->push_back(op);
        break;
      case QHBM_GATE_XPOW:
        op.type = LOW_MAT1;
        ops->push_back(op);
        break;
      case QHBM_GATE_CNOTPOW:
        if (lean_clifford) {  // CNOT**t = H_t CZ**t H_t (X**t = H Z**t H): the controlled phase is the gate's op
          push_hadamard(op.b1, op.gate, 1.f, true);
          LoweredOp cz = op;
          cz.kind = QHBM_GATE_CZPOW;
          cz.type = LOW_DIAG;  // exp(i pi t) on |11>
          ops->push_back(cz);
          push_hadamard(op.b1, op.gate, 1.f, true);
          break;
        }
        op.type = LOW_MAT2;
        ops->push_back(op);
        break;
      case QHBM_GATE_XXPOW:
        if (lean_clifford) { push_pair_power(op, 0, 1.f); break; }
        op.type = LOW_MAT2;
        ops->push_back(op);
        break;
      case QHBM_GATE_YYPOW:
        if (lean_clifford) { push_pair_power(op, 1, 1.f); break; }
        op.type = LOW_MAT2;
        ops->push_back(op);
        break;
      case QHBM_GATE_SWAPPOW:
        // SWAP**t = e^{-i pi t / 2} XX**(t/2) YY**(t/2) ZZ**(t/2): the three commute, and on the triplet exactly one of them
        // contributes e^{i pi t / 2}, on the singlet all three
        if (lean_clifford) {
          push_pair_power(op, 0, 0.5f);
          push_pair_power(op, 1, 0.5f);
          push_pair_power(op, 2, 0.5f);
          if (phases) phases->push_back({op.gate, -0.5f});
          break;
        }
        op.type = LOW_MAT2;
        ops->push_back(op);
        break;
      case QHBM_GATE_ISWAPPOW:
        // ISWAP**t = exp(i pi t (XX + YY) / 4) = e^{i pi t / 2} XX**(-t/2) YY**(-t/2)
        if (lean_clifford) {
          push_pair_power(op, 0, -0.5f);
          push_pair_power(op, 1, -0.5f);
          if (phases) phases->push_back({op.gate, 0.5f});
          break;
        }
        op.type = LOW_MAT2;
        ops->push_back(op);
        break;
      default:
        op.type = LOW_MAT2;
        ops->push_back(op);
        break;
    }
  }
  // A fixed one-bit phase next to a Z**t on the same bit (nothing non-diagonal on that bit between them) is that
  // gate's exponent plus a constant: Z^(1/2) Z**t = Z**(t + 1/2).  In an ansatz of Y**a Z**b layers every S / S^dagger of
  // the Y decompositions disappears this way.
  for (size_t i = 0; i < ops->size();) {
    LoweredOp& f = (*ops)[i];
    if (!(f.fixed && f.type == LOW_DIAG && popc(f.bits) == 1)) { ++i; continue; }
    long host = -1;
    auto usable = [&](const LoweredOp& o) {
      return o.type == LOW_DIAG && o.bits == f.bits && o.mult == 1.f && o.kind == QHBM_GATE_ZPOW && (o.fixed || true);
    };
    for (size_t j = i + 1; j < ops->size(); ++j) {
      const LoweredOp& o = (*ops)[j];
      if (!(o.bits & f.bits)) continue;
      if (o.type != LOW_DIAG) break;
      if (usable(o)) { host = long(j); break; }
    }
    if (host < 0)
      for (size_t j = i; j-- > 0;) {
        const LoweredOp& o = (*ops)[j];
        if (!(o.bits & f.bits)) continue;
        if (o.type != LOW_DIAG) break;
        if (usable(o)) { host = long(j); break; }
      }
    if (host < 0) { ++i; continue; }
    LoweredOp& hst = (*ops)[size_t(host)];
    if (hst.fixed) hst.fixed_t += f.fixed_t;
    else hst.add_offset += f.fixed_t;
    const bool cancelled = hst.fixed && hst.fixed_t == 0.f;  // S^dagger S: nothing left
    if (cancelled) ops->erase(ops->begin() + host);
    ops->erase(ops->begin() + long(i) - (cancelled && size_t(host) < i ? 1 : 0));
    if (cancelled && size_t(host) < i) --i;
  }
  return true;
}

// Ops of `order` (indices into ops) not yet done that a pass with local set S
// can absorb, respecting per-bit order.  Returns them in execution order.
std::vector<int> absorb(const std::vector<LoweredOp>& ops, const std::vector<int>& order,
                        const std::vector<char>& done, uint32_t S, uint32_t all_bits,
                        int* n_mat) {
  std::vector<int> out;
  // Per-bit order is kept between ops that do not commute.  Diagonal ops commute with each other:
  // a diagonal op only waits for a pending NON-diagonal op on one of its bits (blocked_mat), a
  // non-diagonal op for any pending op (blocked_any).  (Treating every pending op as a barrier shrinks
  // the light cone of a chain circuit by two qubits per layer instead of one.)
  uint32_t blocked_any = 0, blocked_mat = 0;
  *n_mat = 0;
  for (int oi : order) {
    if (done[oi]) continue;
    const LoweredOp& op = ops[oi];
    const bool diag = op.type == LOW_DIAG;
    bool ok = false;
    if (!(op.bits & (diag ? blocked_mat : blocked_any))) ok = diag ? (op.bits & S) != 0 : (op.bits & ~S) == 0;
    if (ok) {
      out.push_back(oi);
      if (!diag) ++*n_mat;
    } else {
      blocked_any |= op.bits;
      if (!diag) blocked_mat |= op.bits;
    }
    if ((blocked_mat & all_bits) == all_bits) break;
  }
  return out;
}

struct MeasGroup {
  uint32_t x;
  std::vector<int> terms;  // indices into Model::terms
};

struct Entry {
  uint32_t w0 = 0;  // MAT: mop; CPH: pred
  uint32_t coef_off = 0;
  uint32_t slot = 0xffffffffu;
};

struct Instance {
  uint32_t mat_mask = 0, ph1_mask = 0, ph2_mask = 0, cph_mask = 0;
  uint32_t diag_touch = 0;  // register bits touched by a diagonal entry
  Entry mat[5], ph1[5], ph2[10], cph[10];
};

class Builder {
 public:
  Builder(const Model& m, int K, int R, int n_eff, bool adjoint, Plan* plan)
      : m_(m), K_(K), R_(R), n_eff_(n_eff), adjoint_(adjoint), plan_(plan) {}

  // `phys[b]`: physical position of logical index bit b in the layout the pass loads (null: identity).
  Pass begin_pass(uint32_t S, const std::vector<int>* phys = nullptr) {
    Pass p;
    p.K = K_;
    p.R = R_;
    std::vector<std::pair<int, int>> loc, non;  // (physical position, logical bit)
    p.phys_of.resize(size_t(n_eff_));
    for (int b = 0; b < n_eff_; ++b) {
      const int ph = phys ? (*phys)[size_t(b)] : b;
      p.phys_of[size_t(b)] = ph;
      (S >> b & 1 ? loc : non).push_back({ph, b});
    }
    std::sort(loc.begin(), loc.end());
    std::sort(non.begin(), non.end());
    for (const auto& e : loc) { p.local_phys.push_back(e.first); p.local_pos.push_back(e.second); }
    for (const auto& e : non) { p.nonlocal_phys.push_back(e.first); p.nonlocal_pos.push_back(e.second); }
    int c = 0;
    while (c < K_ && p.local_phys[size_t(c)] == c) ++c;
    p.c = c;
    p.spread.resize(size_t(1) << (K_ - c));
    for (uint32_t j = 0; j < p.spread.size(); ++j) {
      uint32_t v = 0;
      for (int i = c; i < K_; ++i) if (j >> (i - c) & 1) v |= 1u << p.local_phys[size_t(i)];
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

  // Packs the absorbed ops of a pass into rounds / dense two-qubit ops.
  bool emit_ops(Pass* p, const std::vector<LoweredOp>& ops, const std::vector<int>& absorbed,
                std::string* err) {
    std::vector<char> emitted(absorbed.size(), 0);
    size_t left = absorbed.size();
    const uint32_t S = local_set_mask(*p);
    while (left) {
      const size_t before = left;
      // ---- one round: light-cone over up to R register bits -------------------
      uint32_t reg = 0, blocked = 0, blocked_mat = 0;  // (see absorb(): diagonal ops pass pending diagonal ops)
      std::vector<size_t> seq;
      for (size_t i = 0; i < absorbed.size(); ++i) {
        if (emitted[i]) continue;
        const LoweredOp& op = ops[absorbed[i]];
        const bool diag = op.type == LOW_DIAG;
        if (op.bits & (diag ? blocked_mat : blocked)) {
          blocked |= op.bits;
          if (!diag) blocked_mat |= op.bits;
          continue;
        }
        if (op.type == LOW_MAT1) {
          const uint32_t lb = to_local(*p, op.bits);
          if ((reg & lb) || popc(reg) < R_) { reg |= lb; seq.push_back(i); }
          else { blocked |= op.bits; blocked_mat |= op.bits; }
        } else if (diag) {
          const uint32_t ll = to_local(*p, op.bits & S);
          if (ll & reg) seq.push_back(i);
          else if (popc(reg) < R_) { reg |= ll & (0u - ll); seq.push_back(i); }
          else blocked |= op.bits;
        } else {
          blocked |= op.bits;
          blocked_mat |= op.bits;
        }
      }
      if (!seq.empty() && K_ > R_) {
        // The first-come register set above grows a trapezoid (4, 2 gates deep on a chain); a
        // window of four consecutive local bits whose neighbours are already ahead can hold a
        // diamond (2, 4, 2).  Try every window with the register set FIXED and keep the one that
        // absorbs more one-qubit gates (ties: more ops) than the first-come set.
        auto score = [&](const std::vector<size_t>& q) {
          size_t mats = 0;
          for (size_t i : q) mats += ops[absorbed[i]].type == LOW_MAT1;
          return std::make_pair(mats, q.size());
        };
        auto best = score(seq);
        for (int w = 0; w + R_ <= K_; ++w) {
          const uint32_t fixed = ((1u << R_) - 1u) << w;
          uint32_t blk = 0, blk_mat = 0;
          std::vector<size_t> cand;
          for (size_t i = 0; i < absorbed.size(); ++i) {
            if (emitted[i]) continue;
            const LoweredOp& op = ops[absorbed[i]];
            const bool diag = op.type == LOW_DIAG;
            if (op.bits & (diag ? blk_mat : blk)) {
              blk |= op.bits;
              if (!diag) blk_mat |= op.bits;
              continue;
            }
            if (op.type == LOW_MAT1 && (to_local(*p, op.bits) & fixed)) cand.push_back(i);
            else if (diag && (to_local(*p, op.bits & S) & fixed)) cand.push_back(i);
            else {
              blk |= op.bits;
              if (!diag) blk_mat |= op.bits;
            }
          }
          const auto sc = score(cand);
          if (sc > best) { best = sc; seq.swap(cand); reg = fixed; }
        }
      }
      if (!seq.empty()) {
        for (int i = K_ - 1; i >= 0 && popc(reg) < R_; --i) if (!(reg >> i & 1)) reg |= 1u << i;
        // local bits without a pending non-diagonal op at the START of this round (adjoint: psi is
        // back to the input bit there)
        uint32_t pending = pending_mat_outside_;
        for (size_t i = 0; i < absorbed.size(); ++i)
          if (!emitted[i] && ops[absorbed[i]].type != LOW_DIAG) pending |= ops[absorbed[i]].bits;
        round_finished_local_ = adjoint_ ? to_local(*p, ~pending & S) : 0u;
        emit_round(p, ops, absorbed, seq, reg, S);
        for (size_t i : seq) { emitted[i] = 1; --left; }
      }
      // ---- dense two-qubit gates that are ready (applied directly on LDS) ---------
      uint32_t blocked2 = 0;
      for (size_t i = 0; i < absorbed.size(); ++i) {
        if (emitted[i]) continue;
        const LoweredOp& op = ops[absorbed[i]];
        if (op.bits & blocked2) { blocked2 |= op.bits; continue; }
        if (op.type == LOW_MAT2) { emit_gate2(p, op); emitted[i] = 1; --left; }
        else blocked2 |= op.bits;
      }
      if (left == before) { *err = "internal: round packing made no progress"; return false; }
    }
    // Barrier elision.  A thread's amplitudes are those whose non-register local bits spell its
    // id, so a WAVE owns the amplitudes whose wave bits (emit_round: round_wavemasks) spell the wave
    // index.  When two consecutive rounds use the same wave bits, every wave reads back only what it
    // wrote itself (LDS serves one wave's accesses in order): no workgroup barrier between them.
    const int wave_bits = K_ - R_ - 6;  // log2(waves per workgroup)
    for (size_t r = 0; r + 1 < p->round_words.size(); ++r) {
      if (p->round_words[r + 1] != p->round_words[r] + kRoundWords) continue;  // something else sits in between
      if (wave_bits <= 0 || p->round_wavemasks[r] == p->round_wavemasks[r + 1])
        p->prog[p->round_words[r]] |= kRoundNoBarrier;
    }
    return true;
  }

  // Allocates the gradient slot of a parametrised micro-op; returns its index LOCAL to the pass
  // (what the records hold: the kernel's LDS cell and tile_grad column).  `scale` is the factor
  // between the kernel's raw partial Im<lam|A|psi> and dE/dt of that op class: pi for the X / Y
  // involutions, -2 pi for diagonal terms, 1 where the staged generator already carries it.
  int new_slot(Pass* p, const LoweredOp& op, float scale) {
    if (!adjoint_) return -1;
    const Gate& G = m_.gates[op.gate];
    if (op.fixed || G.param_idx < 0 || m_.frozen(G.param_idx)) return -1;
    const int slot = static_cast<int>(plan_->slot_gate.size());
    plan_->slot_gate.push_back(op.gate);
    plan_->slot_factor.push_back(scale * G.scalar * (op.type == LOW_DIAG ? op.mult : 1.f));
    ++p->n_slots;
    return slot - p->slot_base;
  }

  // set by build_plan / emit_ops for the round being emitted
  uint32_t pending_mat_outside_ = 0;   // index bits of non-diagonal ops that are neither done nor in this pass
  uint32_t round_finished_local_ = 0;  // adjoint: local bits with no non-diagonal op left at the round's start

  CoefJob base_job(const LoweredOp& op) {
    const Gate& G = m_.gates[op.gate];
    CoefJob job{};
    job.op_kind = op.kind;
    job.gate = op.gate;
    job.param_idx = G.param_idx;
    job.scalar = G.scalar;
    job.offset = G.offset + op.add_offset;
    if (op.fixed) {  // a constant of the decomposition: not the gate's exponent, never shifted with it
      job.gate = -2;
      job.param_idx = -1;
      job.scalar = 0.f;
      job.offset = op.fixed_t;
    }
    job.out_off = 0;
    job.dagger = adjoint_ ? 1 : 0;
    job.mult = 1.f;
    return job;
  }

  uint32_t alloc_coef(int words, int align) {
    size_t base = (plan_->coef_init.size() + size_t(align) - 1) / size_t(align) * size_t(align);
    plan_->coef_init.resize(base + size_t(words), 0u);
    plan_->n_coef_floats = int(plan_->coef_init.size());
    return uint32_t(base);
  }

  void emit_round(Pass* p, const std::vector<LoweredOp>& ops, const std::vector<int>& absorbed,
                  const std::vector<size_t>& seq, uint32_t reg, uint32_t S) {
    auto rank_of = [&](uint32_t local_bit_mask) { return popc(reg & (local_bit_mask - 1)); };
    struct Placed { int inst; int kind; int j, j2, k; uint32_t pred; const LoweredOp* op; };
    // kind: 0 X, 1 Y, 2 dense, 3 PH1, 4 PH2, 5 CPH
    std::vector<Instance> insts(1);
    std::vector<Placed> placed;
    // An op goes into the EARLIEST instance it may legally run in, not only the last one: ops on different register
    // bits commute, and so do diagonal ops among themselves.  Inside an instance the kernels run the one-qubit gates
    // first and the diagonal terms after them (forward), or the other way round (adjoint), so per register bit:
    //   a gate must come after the last gate on its bit, and after (forward) / not before (adjoint) the last
    //   diagonal term that touches it; a diagonal term must come not before (forward) / after (adjoint) the last
    //   gate on each of its register bits.
    // An inverted circuit lists a layer as Z_q^-1 X_q^-1 per qubit: in sweep order X_a Z_a X_b Z_b ..., which the
    // last-instance rule packed as ONE gate per instance (U_model^dagger of a QMHL step: 320 instances for 320 gates).
    int last_mat[8], last_diag[8];
    for (int j = 0; j < 8; ++j) last_mat[j] = last_diag[j] = -1;
    const bool pack = !std::getenv("QHBM_NO_INSTANCE_PACKING");
    auto instance_at = [&](int idx) -> Instance* {
      while (int(insts.size()) <= idx) insts.emplace_back();
      return &insts[size_t(idx)];
    };
    for (size_t i : seq) {
      const LoweredOp& op = ops[absorbed[i]];
      Placed pl{};
      pl.op = &op;
      if (op.type == LOW_MAT1) {
        const int j = rank_of(to_local(*p, op.bits));
        int at = std::max(last_mat[j] + 1, adjoint_ ? last_diag[j] : last_diag[j] + 1);
        if (!pack) at = std::max(at, int(insts.size()) - 1);
        at = std::max(at, 0);
        Instance* in = instance_at(at);
        last_mat[j] = at;
        pl.inst = at;
        in->mat_mask |= 1u << j;
        pl.kind = op.kind == QHBM_GATE_XPOW ? 0 : (op.kind == QHBM_GATE_YPOW ? 1 : 2);
        if (pl.kind != 0) p->flags |= PASS_GENERAL;
        in->mat[j].w0 = uint32_t(pl.kind);
        pl.j = j;
        ++p->n_mat_ops;
        p->mat_bits |= op.bits;
      } else {  // diagonal term
        const uint32_t ll = to_local(*p, op.bits & S);
        const uint32_t in_reg = ll & reg;
        const uint32_t other_local = ll & ~reg;
        const uint32_t other_nonlocal = op.bits & ~S;
        const int j = rank_of(in_reg & (0u - in_reg));
        int j2 = -1;
        if (popc(in_reg) == 2) { pl.kind = 4; j2 = rank_of(in_reg & (in_reg - 1)); }
        else if (other_local) { pl.kind = 5; pl.pred = uint32_t(__builtin_ctz(other_local)); }
        else if (other_nonlocal) {  // a tile bit: bit K + i of the kernels' index word = bit i of the tile id (kernels.hip cph_*)
          const int b = __builtin_ctz(other_nonlocal);
          const size_t i = size_t(std::find(p->nonlocal_pos.begin(), p->nonlocal_pos.end(), b) - p->nonlocal_pos.begin());
          pl.kind = 5;
          pl.pred = uint32_t(K_ + int(i)) | (1u << 8);
        }
        else pl.kind = 3;
        const uint32_t touch = (1u << j) | (j2 >= 0 ? (1u << j2) : 0u);
        auto fits = [&](const Instance& in) {
          if (pl.kind == 3) return !(in.ph1_mask >> j & 1);
          if (pl.kind == 4) return !(in.ph2_mask >> pair_index(j, j2) & 1);
          return ((in.cph_mask >> (2 * j)) & 3u) != 3u;
        };
        int at = adjoint_ ? last_mat[j] + 1 : last_mat[j];
        if (j2 >= 0) at = std::max(at, adjoint_ ? last_mat[j2] + 1 : last_mat[j2]);
        if (!pack) at = std::max(at, int(insts.size()) - 1);
        at = std::max(at, 0);
        while (at < int(insts.size()) && !fits(insts[size_t(at)])) ++at;
        Instance* in = instance_at(at);
        last_diag[j] = std::max(last_diag[j], at);
        if (j2 >= 0) last_diag[j2] = std::max(last_diag[j2], at);
        pl.inst = at;
        in->diag_touch |= touch;
        pl.j = j;
        pl.j2 = j2;
        if (pl.kind == 3) in->ph1_mask |= 1u << j;
        else if (pl.kind == 4) in->ph2_mask |= 1u << pair_index(j, j2);
        else {
          pl.k = 2 * j + (((in->cph_mask >> (2 * j)) & 1u) ? 1 : 0);
          in->cph_mask |= 1u << pl.k;
        }
        ++p->n_diag_terms;
      }
      placed.push_back(pl);
    }
    // ---- fixed-layout records (program.h RecordLayout), consecutive in the coefficient buffer
    const RecordLayout L(R_, adjoint_);
    const uint32_t first = alloc_coef(L.words() * int(insts.size()), 64);
    for (size_t ii = 0; ii < insts.size(); ++ii) {
      const Instance& in = insts[ii];
      uint32_t kmask[3] = {0, 0, 0};
      for (int j = 0; j < R_; ++j) if (in.mat_mask >> j & 1) kmask[in.mat[j].w0] |= 1u << j;
      uint32_t* rec = &plan_->coef_init[first + ii * size_t(L.words())];
      rec[0] = kmask[0] | (in.ph1_mask << 8) | (in.ph2_mask << 16);
      rec[1] = in.cph_mask | (kmask[1] << 16) | (kmask[2] << 24);
      // one table multiply per amplitude (15 x 4 issue slots) beats per-term phases
      // (18 per PH1, 10 per PH2) once the instance holds enough of them
      if (kmask[1] == 0 && 18 * popc(in.ph1_mask) + 10 * popc(in.ph2_mask) > plan_->full_threshold) {
        rec[0] = kmask[0] | (in.ph1_mask << 4) | (in.ph2_mask << 24);
        rec[1] |= kFullDiagFlag;
      }
      plan_->record_offsets.push_back(uint32_t(first + ii * size_t(L.words())));
      if (adjoint_) for (int k = 0; k < 64; ++k) rec[L.slot0() + k] = 0xffffffffu;
    }
    for (const Placed& pl : placed) {
      const LoweredOp& op = *pl.op;
      const uint32_t rb = first + uint32_t(pl.inst) * uint32_t(L.words());
      uint32_t* rec = &plan_->coef_init[rb];
      CoefJob job = base_job(op);
      int lane = 0, slot_lane = 0;
      switch (pl.kind) {
        case 0: job.mop = MOP_X; lane = L.x(pl.j); slot_lane = L.slot_x(pl.j); break;
        case 1: job.mop = MOP_Y; lane = L.y(pl.j); slot_lane = L.slot_y(pl.j); break;
        case 2: job.mop = MOP_MAT1; lane = L.dense(pl.j); slot_lane = L.slot_dense(pl.j); break;
        case 3:
          job.mop = MOP_PHASE;
          lane = (rec[1] & kFullDiagFlag) ? L.in_ph1(pl.j) : L.ph1(pl.j);
          slot_lane = L.slot_ph1(pl.j);
          break;
        case 4:
          job.mop = MOP_PHASE;
          lane = (rec[1] & kFullDiagFlag) ? L.in_ph2(pair_index(pl.j, pl.j2)) : L.ph2(pair_index(pl.j, pl.j2));
          slot_lane = L.slot_ph2(pair_index(pl.j, pl.j2));
          break;
        default: job.mop = MOP_PHASE; lane = L.cph(pl.k); slot_lane = L.slot_cph(pl.k); rec[L.pred(pl.k)] = pl.pred; break;
      }
      if (job.mop == MOP_PHASE || job.mop == MOP_X) job.mult = op.mult;
      job.out_off = int32_t(rb) + lane;
      plan_->jobs.push_back(job);
      const float kPiF = 3.14159265358979323846f;
      const int slot = new_slot(p, op, pl.kind <= 1 ? kPiF : (pl.kind == 2 ? 1.f : -2.f * kPiF));
      if (adjoint_) rec[slot_lane] = uint32_t(slot);
      if (adjoint_ && pl.kind == 0 && slot >= 0) rec[0] |= 1u << (12 + pl.j);  // X with a gradient slot (program.h word 0)
    }
    p->round_words.push_back(uint32_t(p->prog.size()));
    p->prog.push_back(OP_ROUND | (uint32_t(insts.size()) << 8));
    p->prog.push_back(reg);
    p->prog.push_back(first);
    p->prog.push_back(uint32_t(p->round_tl.size()));
    const size_t dead_word = p->prog.size();
    p->prog.push_back(0u);  // dead mask, set below
    {  // TL[tid]: the thread's local index.  Lane bits (tid 0..5) go to the lowest free local bits,
      // wave bits (tid 6..) to the rest.  A boundary controlled phase is predicated on a free local
      // bit; when that bit spells the WAVE index the predicate is wave-uniform and the kernels skip
      // the whole micro-op in the waves where it is off (cph_fwd / cph_adj) -- so the bits that this
      // round's predicates use most become its wave bits, the highest free bits fill up.
      const int wave_bits = std::max(0, K_ - R_ - 6);
      int free_pos[16], nf = 0;
      for (int b = 0; b < K_; ++b) if (!(reg >> b & 1)) free_pos[nf++] = b;
      uint32_t wmask = 0;
      if (plan_->cph_wave_bits && wave_bits > 0) {
        // first choice: finished bits (half the waves skip the WHOLE round per such bit)
        for (int j = nf - 1, taken = 0; j >= 0 && taken < wave_bits; --j)
          if (round_finished_local_ >> free_pos[j] & 1u) { wmask |= 1u << free_pos[j]; ++taken; }
        int uses[16] = {0};
        for (const Placed& pl : placed) if (pl.kind == 5 && !(pl.pred >> 8)) ++uses[pl.pred & 15u];
        for (int taken = popc(wmask); taken < wave_bits; ++taken) {
          int best = -1;
          for (int j = nf - 1; j >= 0; --j) {
            const int b = free_pos[j];
            if (wmask >> b & 1u) continue;
            if (uses[b] > 0 && (best < 0 || uses[b] > uses[best])) best = b;
          }
          if (best < 0) break;
          wmask |= 1u << best;
        }
      }
      for (int j = nf - 1, taken = popc(wmask); j >= 0 && taken < wave_bits; --j)
        if (!(wmask >> free_pos[j] & 1u)) { wmask |= 1u << free_pos[j]; ++taken; }
      int order[16], no = 0;  // free local bits in tid-bit order
      for (int j = 0; j < nf; ++j) if (!(wmask >> free_pos[j] & 1u)) order[no++] = free_pos[j];
      for (int j = 0; j < nf; ++j) if (wmask >> free_pos[j] & 1u) order[no++] = free_pos[j];
      for (uint32_t tid = 0; tid < (1u << (K_ - R_)); ++tid) {
        uint32_t tl = 0;
        for (int j = 0; j < nf; ++j) if (tid >> j & 1) tl |= 1u << order[j];
        p->round_tl.push_back(tl);
      }
      p->round_wavemasks.push_back(wmask);
      p->prog[dead_word] = wmask & round_finished_local_;
    }
    p->round_regmasks.push_back(reg);
    ++p->n_rounds;
    p->n_instances += int(insts.size());
  }

  void emit_gate2(Pass* p, const LoweredOp& op) {
    CoefJob job = base_job(op);
    job.mop = MOP_MAT2;
    job.out_off = int32_t(alloc_coef(adjoint_ ? 64 : 32, 4));
    plan_->jobs.push_back(job);
    auto local_bit = [&](int gbit) { return uint32_t(__builtin_ctz(to_local(*p, 1u << gbit))); };
    const int slot = new_slot(p, op, 1.f);
    p->flags |= PASS_GENERAL;
    p->prog.push_back(OP_GATE2 | (uint32_t(op.kind) << 8));
    p->prog.push_back(local_bit(op.b0) | (local_bit(op.b1) << 8));
    p->prog.push_back(uint32_t(job.out_off));
    p->prog.push_back(uint32_t(slot));
    ++p->n_mat_ops;
    p->mat_bits |= op.bits;
    ++p->n_rounds;
  }

  void emit_measure(Pass* p, const std::vector<MeasGroup>& groups, const std::vector<int>& all) {
    if (all.empty()) return;
    const uint32_t S = local_set_mask(*p);
    // the diagonal group with many terms goes through the tile's Walsh-Hadamard transform (program.h
    // OP_MEASURE_WHT); its scratch is the top of the kernel's accumulator array: n_ops * 8 + NT * 4 <= 8 * kMaxOps
    std::vector<int> which;
    static_assert(kRoundBits == 4, "OP_MEASURE_WHT has 16 register classes");
    for (int gi : all) {
      const MeasGroup& g = groups[gi];
      // (the PASS's tile, not this builder's: the wide last pass of a forward plan is measured by the plan's builder --
      // classes cut at K_ - R_ = 8 bits for a kernel that cuts them at 9 gave wrong values for every plan whose
      // measuring pass is the wide one, e.g. 38 Z-string shards at 19 qubits)
      const int Kp = p->K;
      const bool wht = g.x == 0 && g.terms.size() >= kWhtMinTerms && R_ == kRoundBits && Kp - R_ >= 6 &&
                       size_t(m_.n_ops) * 8 + (size_t(4) << (Kp - R_)) <= size_t(8) * kMaxOps && !std::getenv("QHBM_NO_WHT");
      if (!wht) { which.push_back(gi); continue; }
      std::vector<std::pair<uint32_t, int>> order;  // (class, term)
      for (int ti : g.terms) order.push_back({to_local(*p, m_.terms[ti].z & S) >> (Kp - R_), ti});
      std::stable_sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
      p->prog.push_back(OP_MEASURE_WHT | (uint32_t(order.size()) << 8));
      for (uint32_t c = 0; c < 16u; ++c) {
        uint32_t end = 0;
        for (const auto& e : order) end += e.first <= c;
        p->prog.push_back(end);
      }
      for (const auto& e : order) {
        const PauliTerm& t = m_.terms[e.second];
        uint32_t cb;
        std::memcpy(&cb, &t.coeff, 4);
        p->prog.push_back(to_local(*p, t.z & S));
        p->prog.push_back(t.z & ~S);
        p->prog.push_back(cb);
        p->prog.push_back(uint32_t(t.op));
        ++p->n_meas_terms;
      }
      ++p->n_meas_groups;
    }
    if (which.empty()) return;
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

bool build_plan(const Model& m, int tile_bits, int round_bits, bool adjoint, Plan* plan, std::string* err,
                int full_threshold, int meas_tile_bits, bool cph_wave_bits, bool relabel, int wide_last_pass,
                const std::vector<uint32_t>* forced_order) {
  *plan = Plan();
  plan->full_threshold = full_threshold;
  plan->cph_wave_bits = cph_wave_bits;
  plan->tail_tiles = cph_wave_bits;  // (one developer switch for the scheduler's round-2 layout choices)
  if (m.n < 1 || m.n > kMaxQubits - 1) { *err = "n_qubits must be in [1, 31]"; return false; }
  const int n_eff = std::max(m.n, kMinTileBits);
  const int k_cap = adjoint ? kMaxTileBits - 1 : kMaxTileBits;
  int K;
  if (tile_bits == 0) {
    // Tiles of 2^12 amplitudes (four 256-thread workgroups per CU) are best for the adjoint sweep (the
    // engine tries 2^13 against the model, engine.cpp) and for the forward sweep up to 21 qubits; from 22
    // on the forward's extra passes cost more than the smaller tiles save.  XXZ chain, depth 16, forward ms
    // with 2^12 / 2^13 (round 4): 20 qubits 97.2 / 123.4 per 4096 states, 21: 109.4 / 108.9 per 2048,
    // 22: 7.1 / 6.7 per 64, 23: 16.0 / 15.7, 24 (config 4's lean passes): 2793 / 2639 per 4546 programs.
    K = n_eff <= k_cap ? n_eff : (adjoint || n_eff <= 21 ? 12 : 13);
  } else {
    if (tile_bits < kMinTileBits || tile_bits > k_cap) {
      *err = "tile_qubits out of range";
      return false;
    }
    K = std::min(n_eff, tile_bits);
  }
  const int R = kRoundBits;
  if (round_bits != 0 && round_bits != kRoundBits) {
    *err = "round_qubits must be 4";
    return false;
  }
  plan->n = m.n;
  plan->n_eff = n_eff;
  plan->K = K;
  plan->R = R;
  plan->adjoint = adjoint;

  std::vector<LoweredOp> ops;
  if (!lower(m, &ops, err, &plan->const_phase, &plan->gate_phases)) return false;
  // Frozen parameters (Model::param_frozen): the backward sweep un-applies the circuit from its end and may stop
  // at the first gate, in circuit order, whose parameter wants a gradient -- everything before it only moves
  // (psi, lambda) further back for nobody.  What is left of psi there is not a basis state: every index bit
  // stays "pending" for good (`always_pending`), so no tile, wave or line is ever pruned.
  uint32_t always_pending = 0;
  if (adjoint && !m.param_frozen.empty() && m.stop_at_first_live_gate) {
    size_t first = 0;
    while (first < ops.size() && (m.gates[size_t(ops[first].gate)].param_idx < 0 || m.frozen(m.gates[size_t(ops[first].gate)].param_idx))) ++first;
    bool mixes = false;  // (dropping diagonal gates alone leaves psi a basis state times a phase)
    for (size_t i = 0; i < first; ++i) mixes |= ops[i].type != LOW_DIAG;
    if (first > 0) {
      ops.erase(ops.begin(), ops.begin() + long(first));
      if (mixes) { always_pending = (1u << std::max(m.n, kMinTileBits)) - 1u; plan->dense_tail = true; }
    }
  }
  std::vector<int> order(ops.size());
  for (size_t i = 0; i < ops.size(); ++i) order[i] = adjoint ? int(ops.size() - 1 - i) : int(i);

  const uint32_t all_bits = (1u << n_eff) - 1;
  const int c_min = std::min(K, 4);
  Builder b(m, K, R, n_eff, adjoint, plan);
  // Relabeling adjoint plans (schedule.h Pass): the physical position of every logical index bit, and the
  // bits already finished and moved.  A finished bit is moved by the pass that finishes it, within the
  // positions that pass's tile owns (in place: a workgroup still writes only where it read).
  plan->relabel = relabel && adjoint && K < n_eff && plan->tail_tiles;
  uint32_t ever_mat = always_pending;  // bits some non-diagonal op acts on (the others are idle: exact zeros of psi off the input bit)
  for (const LoweredOp& op : ops) if (op.type != LOW_DIAG) ever_mat |= op.bits;
  std::vector<int> phys;
  for (int i = 0; i < n_eff; ++i) phys.push_back(i);
  uint32_t frozen = 0;
  auto pending_mat_of = [&](const std::vector<char>& dn) {
    uint32_t pend = always_pending;
    for (size_t oi = 0; oi < ops.size(); ++oi) if (!dn[oi] && ops[oi].type != LOW_DIAG) pend |= ops[oi].bits;
    return pend;
  };
  // The layout after a pass with local set S has run to the state `dn_after`: the bits of S that have no
  // non-diagonal op left (and are not moved yet) go to the highest positions the tile owns, the tile's other
  // bits close up below them in their old order.  Returns the newly moved bits.
  auto relabel_after = [&](uint32_t S, const std::vector<char>& dn_after, std::vector<int>* ph, uint32_t* frz) {
    const uint32_t f_new = S & all_bits & ~pending_mat_of(dn_after) & ~*frz & ever_mat;
    if (!f_new) return 0u;
    std::vector<std::pair<int, int>> tile;  // (physical position, logical bit)
    for (int bb = 0; bb < n_eff; ++bb) if (S >> bb & 1u) tile.push_back({(*ph)[size_t(bb)], bb});
    std::sort(tile.begin(), tile.end());
    std::vector<int> pos;
    for (const auto& e : tile) pos.push_back(e.first);
    size_t next = 0;
    for (const auto& e : tile) if (!(f_new >> e.second & 1u)) (*ph)[size_t(e.second)] = pos[next++];
    for (const auto& e : tile) if (f_new >> e.second & 1u) (*ph)[size_t(e.second)] = pos[next++];
    *frz |= f_new;
    return f_new;
  };
  std::vector<char> done(ops.size(), 0);
  std::vector<int> op_pass(ops.size(), 0);  // pass that executes each lowered op
  size_t n_done = 0;

  // Measurement groups: terms of equal X-mask share conj(psi[l ^ x]) psi[l].
  std::vector<MeasGroup> groups;
  if (!adjoint) {
    for (size_t ti = 0; ti < m.terms.size(); ++ti) {
      const uint32_t x = m.terms[ti].x;
      auto it = std::find_if(groups.begin(), groups.end(), [&](const MeasGroup& g) { return g.x == x; });
      if (it == groups.end()) { groups.push_back(MeasGroup{x, {}}); it = groups.end() - 1; }
      it->terms.push_back(int(ti));
    }
  }
  // First pass after which group gi may be measured: every op that does not commute with one of
  // its terms has run (ops on disjoint bits commute; so does a diagonal op that meets the term
  // only where the term is Z).  Ops not yet scheduled count as running in pass `pending_pass`.
  auto group_ready = [&](size_t gi, int pending_pass) {
    int ready = 0;
    for (int ti : groups[gi].terms) {
      const uint32_t supp = m.terms[size_t(ti)].x | m.terms[size_t(ti)].z;
      for (size_t oi = 0; oi < ops.size(); ++oi) {
        if (!(ops[oi].bits & supp)) continue;
        if (ops[oi].type == LOW_DIAG && !(ops[oi].bits & m.terms[size_t(ti)].x)) continue;
        ready = std::max(ready, done[oi] ? op_pass[oi] : pending_pass);
      }
    }
    return ready;
  };
  auto pass_set = [&](const Pass& q) {
    uint32_t S = 0;
    for (int bb : q.local_pos) S |= 1u << bb;
    return S;
  };

  // Candidate local sets of the next pass, given the ops already done (and, in relabeling plans, the
  // physical layout `ph`: the tiles are built from the LIVE bits -- those with a non-diagonal op left --
  // in the order of their physical positions, the four lowest of which are in every tile).
  auto gen_cands = [&](const std::vector<char>& dn, const std::vector<int>& ph) {
    // ---- candidate local sets -------------------------------------------------
    std::vector<uint32_t> cands;
    if (K >= n_eff) {
      cands.push_back(all_bits);
    } else if (plan->relabel) {
      const uint32_t pending_mat = pending_mat_of(dn);
      std::vector<int> live, rest;  // by physical position
      {
        std::vector<std::pair<int, int>> a, r;
        for (int bit = 0; bit < n_eff; ++bit) (pending_mat >> bit & 1u ? a : r).push_back({ph[size_t(bit)], bit});
        std::sort(a.begin(), a.end());
        std::sort(r.begin(), r.end());
        for (const auto& e : a) live.push_back(e.second);
        for (const auto& e : r) rest.push_back(e.second);
      }
      auto fill = [&](uint32_t S) {  // live bits first (highest position first, as the plain plans fill), then the others
        for (size_t i = live.size(); i-- > 0 && popc(S) < K;) S |= 1u << live[i];
        for (size_t i = 0; i < rest.size() && popc(S) < K; ++i) S |= 1u << rest[i];
        return S;
      };
      uint32_t low = 0;
      for (size_t i = 0; i < live.size() && int(i) < c_min; ++i) low |= 1u << live[i];
      if (int(live.size()) <= K) {
        cands.push_back(fill(0u));
      } else {
        const size_t h = size_t(K - c_min);
        for (size_t p0 = size_t(c_min); p0 + h <= live.size(); ++p0) {
          uint32_t S = low;
          for (size_t k = 0; k < h; ++k) S |= 1u << live[p0 + k];
          cands.push_back(S);
        }
      }
      // demand-driven: bits of the earliest ready ops (non-diagonal first); a diagonal op asks for its live
      // bits, or for one bit if none of them is live any more
      for (int with_diag = 0; with_diag < 2; ++with_diag) {
        uint32_t S = low, blocked = 0;
        for (int oi : order) {
          if (dn[oi]) continue;
          const LoweredOp& op = ops[oi];
          if (op.bits & blocked) { blocked |= op.bits; continue; }
          if (op.type == LOW_DIAG && !with_diag) continue;
          uint32_t need = op.bits;
          if (op.type == LOW_DIAG) {
            if (op.bits & S) continue;
            need = op.bits & pending_mat ? (op.bits & pending_mat) & (0u - (op.bits & pending_mat)) : op.bits & (0u - op.bits);
          }
          if (popc(S | need) <= K) S |= need; else blocked |= op.bits;
        }
        cands.push_back(fill(S));
      }
    } else {
      const uint32_t low = (1u << c_min) - 1;
      const int h = K - c_min;
      for (int p = c_min; p + h <= n_eff; ++p) cands.push_back(low | (((1u << h) - 1) << p));
      // Adjoint tail: once no non-diagonal op is left on one of the low c_min bits, psi is zero wherever
      // it differs from the input bitstring, and a tile that still holds it spends twice the work on
      // zeros.  Tiles over the bits that still have gates (8-byte HBM accesses when index bit 0 is
      // gone: kernels.hip prefetch_tile / store_tile with c == 0) touch the same lines for a half, a
      // quarter, ... a sixteenth of the work.
      if (adjoint && plan->tail_tiles) {
        uint32_t pending_mat = always_pending;
        for (size_t oi = 0; oi < ops.size(); ++oi) if (!dn[oi] && ops[oi].type != LOW_DIAG) pending_mat |= ops[oi].bits;
        const uint32_t low_live = low & pending_mat;  // the low bits that still have gates stay in every tile
        if (low_live != low) {
          const int h2 = K - popc(low_live);
          std::vector<int> ub;
          for (int bit = c_min; bit < n_eff; ++bit) if (pending_mat >> bit & 1u) ub.push_back(bit);
          for (size_t p0 = 0; p0 + size_t(h2) <= ub.size(); ++p0) {
            uint32_t S = low_live;
            for (int k = 0; k < h2; ++k) S |= 1u << ub[p0 + size_t(k)];
            cands.push_back(S);
          }
        }
      }
      // demand-driven: bits of the earliest ready ops (non-diagonal first)
      for (int with_diag = 0; with_diag < 2; ++with_diag) {
        uint32_t S = low, blocked = 0;
        for (int oi : order) {
          if (dn[oi]) continue;
          const LoweredOp& op = ops[oi];
          if (op.bits & blocked) { blocked |= op.bits; continue; }
          if (op.type == LOW_DIAG && !with_diag) continue;
          if (popc(S | op.bits) <= K) S |= op.bits; else blocked |= op.bits;
        }
        for (int bit = n_eff - 1; bit >= 0 && popc(S) < K; --bit) if (!(S >> bit & 1)) S |= 1u << bit;
        cands.push_back(S);
      }
    }
    return cands;
  };
  // What a pass with local set S absorbs (adjoint: cut where its gradient slots run out).
  auto absorb_capped = [&](const std::vector<char>& dn, uint32_t S, int* n_mat) {
    std::vector<int> lst = absorb(ops, order, dn, S, all_bits, n_mat);
    if (adjoint) {  // bound the gradient slots one pass owns (LDS accumulators)
      size_t keep = 0;
      int slots = 0;
      for (; keep < lst.size(); ++keep)
        if (m.gates[ops[lst[keep]].gate].param_idx >= 0 && !m.frozen(m.gates[ops[lst[keep]].gate].param_idx) && ++slots > kMaxSlotsPerPass) break;
      if (keep < lst.size()) {
        lst.resize(keep);
        *n_mat = 0;
        for (int oi : lst) *n_mat += ops[size_t(oi)].type != LOW_DIAG;
      }
    }
    return lst;
  };

  // Adjoint plans: the ORDER of the passes is searched, not grown greedily.  A pass costs about
  // (fixed + gates) x (share of its tiles that is not pruned), at least the HBM time of the lines it
  // touches, and tiles are pruned on every finished bit outside the tile (engine.cpp zero_mask, tail tiles): getting the first bits finished early
  // -- by smaller passes that build the staircase the low bits need -- makes every later gate several
  // times cheaper.  Beam search over pass sequences on that model (config 3: the first four, unpruned
  // passes carry 184 gates instead of 220).
  std::vector<uint32_t> planned;  // local sets of the passes, in order (empty: greedy)
  std::vector<std::pair<double, std::vector<uint32_t>>> complete;  // every complete order the search reached
  if (forced_order) planned = *forced_order;
  if (adjoint && K < n_eff && plan->tail_tiles && !forced_order) {
    struct Node { std::vector<char> dn; size_t n_done; double cost; std::vector<uint32_t> sets; std::vector<int> phys; uint32_t frozen; };
    double kFixed = 6.0, kMemory = 17.0;
    if (const char* e = std::getenv("QHBM_PLAN_KFIXED")) kFixed = std::atof(e);    // developer knobs (scripts/plan_constants_probe.sh)
    if (const char* e = std::getenv("QHBM_PLAN_KMEMORY")) kMemory = std::atof(e);  // per pass, in units of one unpruned gate (config 3: 24.6 ms of tile I/O against 1.4 ms per gate; the fixed part 3, 5 and 7 measured alike, 10 and 16 worse)
    size_t kBeam = ops.size() <= 4000 ? 16 : (ops.size() <= 12000 ? 8 : 4);  // planning time stays well below a second
    if (const char* e = std::getenv("QHBM_PLAN_BEAM")) kBeam = size_t(std::max(1, std::atoi(e)));
    auto finished_bits = [&](const std::vector<char>& dn) {
      uint32_t pend = always_pending;
      for (size_t oi = 0; oi < ops.size(); ++oi) if (!dn[oi] && ops[oi].type != LOW_DIAG) pend |= ops[oi].bits;
      return all_bits & ~pend;
    };
    std::vector<Node> beam(1);
    beam[0].dn = done;
    beam[0].n_done = n_done;
    beam[0].cost = 0.0;
    beam[0].phys = phys;
    beam[0].frozen = frozen;
    double best_cost = -1.0;
    for (int depth = 0; depth < 256 && !beam.empty(); ++depth) {
      std::vector<Node> next;
      for (const Node& nd : beam) {
        const uint32_t fin = finished_bits(nd.dn);
        std::vector<uint32_t> cs = gen_cands(nd.dn, nd.phys);
        std::sort(cs.begin(), cs.end());
        cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
        for (uint32_t S : cs) {
          int n_mat = 0;
          std::vector<int> lst = absorb_capped(nd.dn, S, &n_mat);
          if (lst.empty()) continue;
          Node c;
          c.dn = nd.dn;
          for (int oi : lst) c.dn[size_t(oi)] = 1;
          c.n_done = nd.n_done + lst.size();
          // (fixed + gates) on the tiles that are not pruned, but never less than the HBM time of the
          // 128-byte lines those tiles touch: a tile without k of the four low index bits uses 2^-k of a line
          const double alive = 1.0 / double(1u << std::min(20, popc(fin & ~S)));
          // (relabeling plans keep finished bits out of the lines: the live tiles are made of whole lines)
          const double lines = plan->relabel ? alive : std::min(1.0, alive * double(1u << popc(((1u << c_min) - 1u) & ~S)));
          c.cost = nd.cost + std::max(kMemory * lines, alive * (kFixed + double(n_mat)));
          c.sets = nd.sets;
          c.sets.push_back(S);
          c.phys = nd.phys;
          c.frozen = nd.frozen;
          if (plan->relabel && c.n_done < ops.size()) relabel_after(S, c.dn, &c.phys, &c.frozen);
          if (c.n_done == ops.size()) {
            complete.emplace_back(c.cost, c.sets);
            if (best_cost < 0.0 || c.cost < best_cost) {
              best_cost = c.cost;
              planned = c.sets;
              if (std::getenv("QHBM_PLAN_DEBUG")) {
                std::fprintf(stderr, "[plan] complete at depth %d cost %.2f:", depth, c.cost);
                for (uint32_t st : c.sets) std::fprintf(stderr, " %x", st);
                std::fprintf(stderr, "\n");
              }
            }
            continue;
          }
          bool dup = false;
          for (Node& o : next)
            if (o.n_done == c.n_done && o.dn == c.dn) { dup = true; if (c.cost < o.cost) o = c; break; }
          if (!dup) next.push_back(std::move(c));
        }
      }
      auto rank = [&](const Node& nd) {
        size_t mats_left = 0;
        for (size_t oi = 0; oi < ops.size(); ++oi) mats_left += !nd.dn[oi] && ops[oi].type != LOW_DIAG;
        const int nfin = std::min(4, popc(finished_bits(nd.dn) & ((1u << m.n) - 1u)));
        return nd.cost + double(mats_left) / double(1 << nfin);
      };
      std::sort(next.begin(), next.end(), [&](const Node& x, const Node& y) { return rank(x) < rank(y); });
      if (std::getenv("QHBM_PLAN_DEBUG"))
        for (size_t i = 0; i < next.size() && i < 20; ++i) {
          std::fprintf(stderr, "[plan] depth %d #%zu cost %.2f rank %.2f done %zu frozen %x:", depth, i, next[i].cost, rank(next[i]), next[i].n_done, next[i].frozen);
          for (uint32_t st : next[i].sets) std::fprintf(stderr, " %x", st);
          std::fprintf(stderr, "\n");
        }
      if (next.size() > kBeam) next.resize(kBeam);
      if (best_cost >= 0.0) {  // drop what cannot beat the best complete sequence
        next.erase(std::remove_if(next.begin(), next.end(), [&](const Node& nd) { return nd.cost >= best_cost; }), next.end());
      }
      beam.swap(next);
    }
  }
  if (!planned.empty() && !forced_order) {
    // The model knows nothing of how well a pass packs into rounds and instances: a searched order with
    // clearly MORE passes than the greedy one (deep circuits on many qubits, where no bit finishes
    // early anyway: config 5, 24 against 21) measured slower, so the greedy order stands there.
    std::vector<char> dn(done);
    std::vector<int> gphys(phys);
    uint32_t gfrozen = frozen;
    size_t left = ops.size() - n_done, greedy_passes = 0;
    std::vector<uint32_t> greedy_sets;
    while (left) {
      int best_mat = -1;
      size_t best_total = 0;
      uint32_t best_set = 0;
      std::vector<int> best;
      for (uint32_t S : gen_cands(dn, gphys)) {
        int n_mat = 0;
        std::vector<int> lst = absorb_capped(dn, S, &n_mat);
        if (n_mat > best_mat || (n_mat == best_mat && lst.size() > best_total)) { best_mat = n_mat; best_total = lst.size(); best_set = S; best.swap(lst); }
      }
      if (best.empty()) break;
      for (int oi : best) dn[size_t(oi)] = 1;
      left -= best.size();
      ++greedy_passes;
      greedy_sets.push_back(best_set);
      if (plan->relabel && left) relabel_after(best_set, dn, &gphys, &gfrozen);
    }
    // (relabeling plans: a pruned pass moves only its live lines, so more, smaller tail passes are cheap -- the
    // searched order stands unless the greedy one has FAR fewer passes)
    // the candidates handed to the caller: the searched orders by proxy cost (the chosen one excluded), the greedy one
    std::sort(complete.begin(), complete.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    for (const auto& c : complete) {
      if (plan->candidate_orders.size() >= 6) break;
      if (c.second != planned) plan->candidate_orders.push_back(c.second);
    }
    if (left == 0) plan->candidate_orders.push_back(greedy_sets);
    if (left == 0 && planned.size() > greedy_passes + (plan->relabel ? 4 : 1)) planned.clear();
  }
  size_t planned_i = 0;
  {  // developer knobs: QHBM_ADJ_SETS / QHBM_FWD_SETS=hex,hex,... force the local sets of the first passes
    if (const char* env = std::getenv(adjoint ? "QHBM_ADJ_SETS" : "QHBM_FWD_SETS")) {
      planned.clear();
      for (const char* q = env; *q;) {
        char* end = nullptr;
        const unsigned long v = std::strtoul(q, &end, 16);
        if (end == q) break;
        planned.push_back(uint32_t(v));
        q = *end ? end + 1 : end;
      }
    }
  }

  while (n_done < ops.size()) {
    std::vector<uint32_t> cands = planned_i < planned.size() ? std::vector<uint32_t>{planned[planned_i++]} : gen_cands(done, phys);
    // A WIDER last pass: when everything that is left needs one or two index bits more than a tile has, the
    // last gate pass takes a tile of 2^(K+1) or 2^(K+2) amplitudes instead of leaving a pass of its own to a
    // handful of gates (config 3: 33 gates on 13 bits were 32 + 1 in two passes, the second a full read and
    // write of the state for ONE gate: 14 of the forward sweep's 110 ms).
    int wide_K = 0;
    uint32_t wide_S = 0;
    // (not for the first pass, and not against an explicit tile size unless asked for: wide_last_pass = 1)
    const bool widen = wide_last_pass > 0 || (wide_last_pass < 0 && tile_bits == 0);
    if (!adjoint && K < n_eff && plan->tail_tiles && widen && !plan->passes.empty() && !std::getenv("QHBM_NO_WIDE_LAST_PASS")) {
      uint32_t S = (1u << c_min) - 1;
      for (int oi : order) {
        if (done[oi]) continue;
        const LoweredOp& op = ops[oi];
        S |= op.type == LOW_DIAG ? (op.bits & S ? 0u : (op.bits & (0u - op.bits))) : op.bits;
      }
      const int need = popc(S);
      if (need > K && need <= std::min(K + 2, std::min(n_eff, 13))) {  // (13: the paired forward kernel's largest tile)
        wide_K = need;
        wide_S = S;
      }
    }
    if (!adjoint && K < n_eff && !groups.empty()) {
      // If one tile can hold every remaining op, this is the last gate pass: spend its spare local
      // bits on the X-masks of the groups no earlier pass can measure, so that the measurement
      // needs no pass (a full read of the state) of its own -- only if ALL of them fit (a
      // scattered local set that still leaves a measurement pass costs more than it saves).
      // Listed first: ties go to it.
      uint32_t S = (1u << c_min) - 1;
      bool fits = true;
      for (int oi : order) {
        if (done[oi]) continue;
        const LoweredOp& op = ops[oi];
        const uint32_t need = op.type == LOW_DIAG ? (op.bits & S ? 0u : (op.bits & (0u - op.bits))) : op.bits;
        if (popc(S | need) > K) { fits = false; break; }
        S |= need;
      }
      if (fits) {
        const int here = int(plan->passes.size());
        for (size_t gi = 0; gi < groups.size() && fits; ++gi) {
          const int ready = group_ready(gi, here);
          bool earlier = false;
          for (int pi = ready; pi < here && !earlier; ++pi) earlier = (groups[gi].x & ~pass_set(plan->passes[size_t(pi)])) == 0;
          if (earlier) continue;
          if (popc(S | groups[gi].x) <= K) S |= groups[gi].x; else fits = false;  // all or nothing
        }
      }
      if (fits) {
        for (int bit = n_eff - 1; bit >= 0 && popc(S) < K; --bit) if (!(S >> bit & 1)) S |= 1u << bit;
        cands.insert(cands.begin(), S);
      }
    }
    uint32_t best_S = 0;
    int best_mat = -1;
    size_t best_total = 0;
    std::vector<int> best_list;
    // One pass of lookahead: a tile is ranked by the non-diagonal ops it absorbs PLUS the most a
    // following contiguous-block tile could absorb (a pass is a full read and write of the state).
    // Forward plans only: measured on the adjoint plan it saves a pass but costs two rounds (+6 %).
    std::vector<uint32_t> blocks;
    if (K < n_eff) {
      const uint32_t low = (1u << c_min) - 1;
      const int h = K - c_min;
      for (int pp = c_min; pp + h <= n_eff; ++pp) blocks.push_back(low | (((1u << h) - 1) << pp));
    }
    for (uint32_t S : cands) {
      int n_mat = 0;
      std::vector<int> lst = absorb_capped(done, S, &n_mat);
      int next_best = 0;
      if (!adjoint && !blocks.empty() && n_done + lst.size() < ops.size()) {
        std::vector<char> done2(done);
        for (int oi : lst) done2[oi] = 1;
        for (uint32_t S2 : blocks) {
          int m2 = 0;
          absorb(ops, order, done2, S2, all_bits, &m2);
          next_best = std::max(next_best, m2);
        }
      }
      const int score = n_mat + next_best;
      if (score > best_mat || (score == best_mat && lst.size() > best_total)) {
        best_mat = score; best_total = lst.size(); best_S = S; best_list.swap(lst);
      }
    }
    if (best_list.empty()) { *err = "scheduler made no progress"; return false; }
    Builder wide(m, wide_K ? wide_K : K, R, n_eff, adjoint, plan);
    Builder* bp = &b;
    if (wide_K) {  // does the wider tile really take everything that is left?
      int n_mat = 0;
      std::vector<int> all_left = absorb(ops, order, done, wide_S, all_bits, &n_mat);
      if (all_left.size() == ops.size() - n_done) {
        best_S = wide_S;
        best_list.swap(all_left);
        bp = &wide;
      }
    }
    Pass p = bp->begin_pass(best_S, &phys);
    p.frozen_old_local = Builder::to_local(p, frozen & best_S);
    p.slot_base = int(plan->slot_gate.size());
    {
      std::vector<char> here(ops.size(), 0);
      for (int oi : best_list) here[size_t(oi)] = 1;
      uint32_t pend = always_pending;
      for (size_t oi = 0; oi < ops.size(); ++oi)
        if (!done[oi] && !here[oi] && ops[oi].type != LOW_DIAG) pend |= ops[oi].bits;
      bp->pending_mat_outside_ = pend;
    }
    if (!bp->emit_ops(&p, ops, best_list, err)) return false;
    {
      // The zero-tile / dead-wave pruning (engine.cpp fill_args, emit_round's dead masks) is sound only
      // if Pass::mat_bits lists EVERY index bit a non-diagonal op of the pass acts on: a new lowered op
      // kind that mixes the two halves of a bit without reporting it would be pruned wrongly.
      uint32_t mixed = 0;
      for (int oi : best_list) if (ops[size_t(oi)].type != LOW_DIAG) mixed |= ops[size_t(oi)].bits;
      static_assert(LOW_MAT2 == 3, "a new LoweredType must be classified here: diagonal, or reported in mat_bits");
      if (mixed != p.mat_bits) { *err = "internal: a non-diagonal op of the pass is missing from mat_bits"; return false; }
    }
    for (int oi : best_list) { done[oi] = 1; op_pass[oi] = int(plan->passes.size()); ++n_done; }
    if (plan->relabel && n_done < ops.size()) {  // (the last pass stores nothing)
      const std::vector<int> before(phys);
      const uint32_t f_new = relabel_after(best_S, done, &phys, &frozen);
      if (f_new) {
        p.flags |= PASS_RELABEL;
        p.frozen_new_local = Builder::to_local(p, f_new);
        std::vector<int> live_bits;  // local index bits that stay, in ascending physical (= local) order
        for (int i = 0; i < K; ++i) {
          p.store_local_phys.push_back(phys[size_t(p.local_pos[size_t(i)])]);
          if (!(p.frozen_new_local >> i & 1u)) live_bits.push_back(i);
        }
        const uint32_t n_live = uint32_t(live_bits.size());
        p.relabel_tab.resize(size_t(2) << n_live);
        for (uint32_t o = 0; o < (1u << n_live); ++o) {
          uint32_t l = 0, off = 0;
          for (uint32_t j = 0; j < n_live; ++j)
            if (o >> j & 1u) { l |= 1u << live_bits[j]; off |= 1u << p.store_local_phys[size_t(live_bits[j])]; }
          p.relabel_tab[2 * size_t(o)] = l;
          p.relabel_tab[2 * size_t(o) + 1] = off;
        }
      }
    }
    plan->passes.push_back(std::move(p));
  }

  {  // the kernels prefetch one record past the last one of a round
    const RecordLayout L(R, adjoint);
    plan->coef_init.resize(plan->coef_init.size() + size_t(L.words()) + 64, 0u);
    plan->n_coef_floats = int(plan->coef_init.size());
  }
  if (adjoint) {
    if (plan->relabel) {  // the relabeling store exists in the exchange-layout kernel only (lean programs)
      bool general = false;
      for (const Pass& p : plan->passes) general |= (p.flags & PASS_GENERAL) != 0;
      if (general) return build_plan(m, tile_bits, round_bits, adjoint, plan, err, full_threshold, meas_tile_bits, cph_wave_bits, false, wide_last_pass);
    }
    for (Pass& p : plan->passes) {
      p.flags |= PASS_ADJOINT | PASS_STORE;
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
  // The first pass writes the basis state: ONE tile per state is not zero.  If every index bit is acted on
  // by some non-diagonal gate, the other tiles need not be written at all: a later pass skips the tiles that
  // differ from the input on a bit nothing has acted on yet (engine.cpp fill_args: zero_mask), and clears,
  // when it loads a tile, the amplitudes that differ on such a bit among its LOCAL bits -- the only places
  // never written before.  (With an idle bit the final state would keep unwritten regions: zeros are filled.)
  if (plan->passes.size() > 1 && ever_mat == all_bits) {
    plan->passes.front().flags |= PASS_NO_ZERO_FILL;
    uint32_t touched = plan->passes.front().mat_bits;
    for (size_t i = 1; i < plan->passes.size(); ++i) {
      Pass& q = plan->passes[i];
      q.frozen_old_local = Builder::to_local(q, pass_set(q) & ~touched);
      touched |= q.mat_bits;
    }
  }

  std::vector<char> gdone(groups.size(), 0);
  size_t g_left = groups.size();
  // Each X-mask group goes to the earliest pass after its last non-commuting op whose tile holds
  // its flipped bits -- usually a pass that exists anyway.
  {
    std::vector<std::vector<int>> early(plan->passes.size());
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      const int ready = group_ready(gi, int(plan->passes.size()) - 1);
      for (size_t pi = size_t(ready); pi + 1 < plan->passes.size(); ++pi) {  // the last pass is handled below
        if ((groups[gi].x & ~pass_set(plan->passes[pi])) == 0) {
          early[pi].push_back(int(gi));
          gdone[gi] = 1;
          --g_left;
          break;
        }
      }
    }
    for (size_t pi = 0; pi < early.size(); ++pi)
      if (!early[pi].empty()) b.emit_measure(&plan->passes[pi], groups, early[pi]);
  }
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
  // Measurement-only passes (a full read of the state each): the largest tile the forward kernel
  // has (2^14 amplitudes) holds the most X-masks, and the low four index bits stay local so that a
  // tile is made of >= 128-byte contiguous pieces -- with scattered 16-byte pieces (c = 1) these
  // passes ran at a quarter of the HBM rate.  Groups too wide for that get a second chance with
  // only bit 0 pinned; what still does not fit goes to the strided-gather kernel.
  const int K_meas = std::min(n_eff, std::max(K, meas_tile_bits > 0 ? meas_tile_bits : kMaxTileBits));
  Builder bm(m, K_meas, R, n_eff, adjoint, plan);
  for (int c_pin : {c_min, 1}) {
    const uint32_t pinned = (1u << std::min(c_pin, K_meas)) - 1u;
    for (;;) {
      // the x-bits of as many groups as fit (fewest new bits first), then low bits
      uint32_t S = pinned;
      bool any = false;
      for (;;) {
        int best_add = 1 << 30;
        size_t best = groups.size();
        for (size_t gi = 0; gi < groups.size(); ++gi) {
          if (gdone[gi] || (groups[gi].x & ~S) == 0) continue;
          if (popc(S | groups[gi].x) > K_meas) continue;
          const int add = popc(groups[gi].x & ~S);
          if (add < best_add) { best_add = add; best = gi; }
        }
        if (best == groups.size()) break;
        S |= groups[best].x;
        any = true;
      }
      if (!any) {  // nothing left whose bits can be added: either all covered by `pinned`, or too wide
        bool covered = false;
        for (size_t gi = 0; gi < groups.size(); ++gi) covered |= !gdone[gi] && (groups[gi].x & ~S) == 0;
        if (!covered) break;
      }
      for (int bit = 0; bit < n_eff && popc(S) < K_meas; ++bit) if (!(S >> bit & 1)) S |= 1u << bit;
      Pass p = bm.begin_pass(S);
      p.is_measure_only = true;
      const size_t before = g_left;
      take(&p);
      if (g_left == before) break;
      plan->passes.push_back(std::move(p));
    }
    if (!g_left) break;
  }
  // What is left flips more qubits than a tile holds: those terms are measured by the
  // strided-gather kernel on the final state in HBM (kernels.hip measure_global_kernel).
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    if (gdone[gi]) continue;
    for (int ti : groups[gi].terms) plan->global_terms.push_back(ti);
    gdone[gi] = 1;
    --g_left;
  }
  for (Pass& p : plan->passes) p.prog.push_back(OP_END);
  return true;
}

std::string describe_plan(const Plan& p) {
  std::ostringstream os;
  os << (p.adjoint ? (p.relabel ? "adjoint (relabeling)" : "adjoint") : "forward") << " plan: n=" << p.n << " n_eff=" << p.n_eff
     << " tile_bits=" << p.K << " round_bits=" << p.R << " passes=" << p.passes.size()
     << " coef_floats=" << p.n_coef_floats;
  if (!p.global_terms.empty()) os << " global_terms=" << p.global_terms.size();
  os << "\n";
  {  // micro-op census over all instance records (cost model input)
    const RecordLayout L(p.R, p.adjoint);
    int n_inst = 0, n_full = 0, x = 0, ph1 = 0, ph2 = 0, fph1 = 0, fph2 = 0, cph = 0, cph_tile = 0, groups = 0;
    int x_hist[5] = {0, 0, 0, 0, 0};
    auto pc = [](uint32_t v) { return __builtin_popcount(v); };
    for (uint32_t off : p.record_offsets) {
      const uint32_t h0 = p.coef_init[off], h1 = p.coef_init[off + 1];
      ++n_inst;
      x += pc(h0 & 0xfu);
      ++x_hist[pc(h0 & 0xfu)];
      cph += pc(h1 & 0xffu);
      for (int k = 0; k < 8; ++k)
        if ((h1 >> k & 1u) && (p.coef_init[off + L.pred(k)] >> 8)) ++cph_tile;  // predicate on a tile bit: workgroup-uniform
      groups += ((h0 & 0xf0fu) != 0 || ((h1 & kFullDiagFlag) && ((h0 >> 4) & 0xfu))) + ((h1 & 0xffu) != 0);
      if (h1 & kFullDiagFlag) {
        ++n_full;
        fph1 += pc((h0 >> 4) & 0xfu);
        fph2 += pc((h0 >> 24) & 0x3fu);
        groups += ((h0 >> 24) & 0x3fu) != 0;
      } else {
        ph1 += pc((h0 >> 8) & 0xfu);
        ph2 += pc((h0 >> 16) & 0x3fu);
        groups += ((h0 >> 16) & 0x3fu) != 0;
      }
    }
    if (p.adjoint) {  // how full the eight-wide gradient reductions are (kernels.hip add_slots8)
      int hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t off : p.record_offsets)
        for (int g8 = 0; g8 < 4; ++g8) {
          int cnt = 0;
          for (int v = 0; v < 8; ++v) cnt += p.coef_init[off + L.slot_lane8(g8, v)] != 0xffffffffu;
          if (cnt) ++hist[cnt];
        }
      os << "  slots per eight-wide reduction 1..8:";
      for (int c = 1; c <= 8; ++c) os << " " << hist[c];
      int half[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // per slot group: all in values 0..3 / all in 4..7 / both
      for (uint32_t off : p.record_offsets)
        for (int g8 = 0; g8 < 4; ++g8) {
          int lo = 0, hi = 0;
          for (int v = 0; v < 8; ++v) (v < 4 ? lo : hi) += p.coef_init[off + L.slot_lane8(g8, v)] != 0xffffffffu;
          if (lo + hi) ++half[g8][lo && hi ? 2 : (lo ? 0 : 1)];
        }
      os << "; by group (X+PH1 | PH2 | CPH | Y+dense) low-half only / high-half only / both:";
      for (int g8 = 0; g8 < 4; ++g8) os << " " << half[g8][0] << "/" << half[g8][1] << "/" << half[g8][2];
      os << "\n";
    }
    os << "  census: instances=" << n_inst << " (FULL " << n_full << ") X=" << x << " PH1=" << ph1 << " PH2=" << ph2
       << " FULL-PH1=" << fph1 << " FULL-PH2=" << fph2 << " CPH=" << cph << " (tile predicate " << cph_tile << ") slot-groups=" << groups << " instances by X count 0..4: " << x_hist[0] << "/" << x_hist[1] << "/"
       << x_hist[2] << "/" << x_hist[3] << "/" << x_hist[4] << "\n";
  }
  for (size_t i = 0; i < p.passes.size(); ++i) {
    const Pass& q = p.passes[i];
    os << "  pass " << i << ": K=" << q.K << " c=" << q.c << " local=[";
    for (size_t k = 0; k < q.local_pos.size(); ++k) os << (k ? "," : "") << q.local_pos[k];
    os << "] mat_ops=" << q.n_mat_ops << " diag_terms=" << q.n_diag_terms << " rounds=" << q.n_rounds
       << " instances=" << q.n_instances << " meas_groups=" << q.n_meas_groups
       << " meas_terms=" << q.n_meas_terms << " slots=" << q.n_slots
       << (q.is_measure_only ? " [measure-only]" : "") << " words=" << q.prog.size() << " regs=";
    for (size_t r = 0; r < q.round_regmasks.size(); ++r) os << (r ? "," : "") << std::hex << q.round_regmasks[r] << std::dec;
    if (p.relabel) {
      os << " at=";
      for (size_t k = 0; k < q.local_phys.size(); ++k) os << (k ? "," : "") << q.local_phys[k];
      if (q.flags & PASS_RELABEL) os << " moves-local-bits=" << std::hex << q.frozen_new_local << std::dec;
      if (q.frozen_old_local) os << " stale-local-bits=" << std::hex << q.frozen_old_local << std::dec;
    }
    if (p.adjoint) {
      os << " dead=";
      for (size_t r = 0; r < q.round_words.size(); ++r) os << (r ? "," : "") << std::hex << q.prog[q.round_words[r] + 4] << std::dec;
    }
    os << "\n";
  }
  return os.str();
}

}  // namespace qhbm
