// schedule.h -- host-side lowering of a circuit + observables into passes.
//
// Replaces what the reference leaves to TFQ's op: proto parsing, symbol
// resolution and qsim gate fusion per circuit per call
// (/root/reference/qhbmlib/inference/qnn.py:134-138).
// Here the structure is scheduled ONCE per circuit; only the small
// coefficient buffer changes from call to call.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "program.h"

namespace qhbm {

struct Gate {  // layout-compatible with the C ABI's qhbm_gate
  int32_t kind, q0, q1, param_idx;
  float scalar, offset;
  float global_shift;  // cirq's EigenGate shift: a global phase exp(i pi t shift), seen by qhbm_statevector only
};

struct PauliTerm {
  float coeff;
  uint32_t x, z;  // masks in amplitude-index bit space
  int ny;         // number of Y factors
  int op;         // observable this term belongs to
};

struct LoweredOp {
  int type = LOW_SKIP;  // LoweredType
  int kind = 0;         // qhbm_gate_kind
  int gate = -1;        // index into the circuit
  uint32_t bits = 0;    // index bits touched
  int b0 = -1, b1 = -1; // index bit of q0 / q1
  float mult = 1.f;     // LOW_DIAG: phase = exp(i*pi*mult*t) where all `bits` are 1; X**(mult*t) for an X op
  // an op of a gate's DECOMPOSITION with an exponent of its own (Y**t = Z^(1/2) X**t Z^(-1/2): the two phases):
  // no parameter, no gradient slot, never shifted
  bool fixed = false;
  float fixed_t = 0.f;
  float add_offset = 0.f;  // a neighbouring fixed phase on the same bit folded into this Z**t: exponent t + add_offset
};

// One entry of the per-call coefficient preparation.
struct CoefJob {
  int32_t op_kind;    // qhbm_gate_kind
  int32_t mop;        // MOP_X / MOP_Y / MOP_MAT1 / MOP_MAT2 / MOP_PHASE
  int32_t gate;       // circuit gate (for exponent and parameter-shift)
  int32_t param_idx;
  float scalar, offset;
  int32_t out_off;    // float offset in the coefficient buffer
  int32_t swap;       // MAT2: matrix index bits swapped
  int32_t dagger;     // write U^dagger (adjoint plan), followed by the generator
  float mult;         // MOP_PHASE: multiplier of the exponent
};

struct Pass {
  int K = 0, R = 0, c = 0;
  uint32_t flags = 0;
  // LOGICAL index bits (bit n-1-q <-> qubit q) of the tile, size K, and of the rest, size n - K, each in
  // ascending order of their PHYSICAL position (the bit of the amplitude's address in HBM).  Logical and
  // physical positions coincide except in adjoint plans that relabel (Plan::relabel): there a pass that
  // finishes an index bit stores its tiles with that bit moved to the highest position the tile owns, so
  // that in every later pass the live half of the state (finished bit == input bit) is made of WHOLE
  // 128-byte lines instead of every other amplitude of every line.
  std::vector<int> local_pos;
  std::vector<int> nonlocal_pos;
  std::vector<int> local_phys, nonlocal_phys;  // their physical positions when the pass LOADS its tiles (ascending)
  std::vector<int> phys_of;                    // logical bit -> physical position at load time, size n_eff
  std::vector<int> store_local_phys;           // physical position of local index bit i when the pass STORES (empty: unchanged)
  uint32_t frozen_new_local = 0;  // local index bits this pass finishes and moves: stored only where they equal the input bit
  uint32_t frozen_old_local = 0;  // local index bits finished and moved by EARLIER passes: stale data where != input, zeroed at load
  std::vector<uint32_t> relabel_tab;  // relabeling store: per live out-local index o: [2 o] = local index (finished bits clear), [2 o + 1] = out offset
  std::vector<uint32_t> prog;     // instruction words, OP_END terminated
  std::vector<uint32_t> spread;   // spread_hi[2^(K-c)]
  std::vector<uint32_t> round_tl; // TL[tid] tables of the rounds, 2^(K-R) entries each (OP_ROUND word 3)
  bool is_measure_only = false;
  bool completes_circuit = false;
  // statistics (DESIGN.md / bench roofline accounting)
  int n_mat_ops = 0, n_diag_terms = 0, n_rounds = 0, n_instances = 0;
  std::vector<uint32_t> round_regmasks;  // register-bit set of every OP_ROUND (introspection)
  std::vector<uint32_t> round_wavemasks; // local bits that spell the WAVE index in that round (the rest are lane bits)
  std::vector<uint32_t> round_words;     // index in `prog` of every OP_ROUND's first word
  int n_meas_groups = 0, n_meas_terms = 0;
  int slot_base = 0, n_slots = 0;
  uint32_t mat_bits = 0;  // index bits acted on by a non-diagonal op of this pass
};

struct Plan {
  int n = 0, n_eff = 0, K = 0, R = 0;
  bool adjoint = false;
  std::vector<Pass> passes;
  std::vector<CoefJob> jobs;
  int n_coef_floats = 0;
  std::vector<uint32_t> coef_init;  // static words of the coefficient buffer (records)
  std::vector<uint32_t> record_offsets;  // word offset of every instance record
  int full_threshold = 60;  // per-term cost above which an instance uses the FULL diagonal table
  bool tail_tiles = true;     // adjoint tail passes may drop the low index bits once they have no gate left (schedule.cpp)
  bool relabel = false;       // adjoint: finished index bits are moved out of the 128-byte lines (Pass::store_local_phys)
  bool cph_wave_bits = true;  // map the partner bits of boundary controlled phases to wave bits (schedule.cpp emit_round)
  // forward: indices (into Model::terms) of the Pauli terms whose X-mask does not fit a tile; they
  // are measured on the final state in HBM by the strided-gather kernel
  std::vector<int> global_terms;
  // adjoint: gradient slot -> (gate, chain-rule factor to the exponent)
  std::vector<int> slot_gate;
  std::vector<float> slot_factor;
  // adjoint plans: complete pass orders (local sets, in order) the search ranked best by its proxy cost, and the
  // greedy order; the caller may rebuild the plan with any of them (`forced_order`) and keep the one whose flop
  // model is smallest (engine.cpp build_plans)
  std::vector<std::vector<uint32_t>> candidate_orders;
  // adjoint plans of a model with frozen leading gates: the sweep ends at an intermediate state of the circuit, not
  // at the basis state -- no index bit ever "finishes", nothing is pruned (engine.cpp fill_args)
  bool dense_tail = false;
  // constant global phase, in units of pi, that the lowering of constant Hadamards / CNOTs owes the exported state
  // (schedule.cpp lower(): -1/4 per lowered H)
  double const_phase = 0.0;
  // ... and per gate: (gate, f) = a factor e^{i pi f t_gate} (SWAP**t = e^{-i pi t / 2} XX**(t/2) YY**(t/2) ZZ**(t/2))
  std::vector<std::pair<int, float>> gate_phases;
};

struct Model {
  int n = 0;
  int n_params = 0;
  std::vector<Gate> gates;
  // parameters whose gradient nobody asks for (qhbm_set_gradient_mask; empty: none): their gates get no gradient
  // slot, and the backward sweep stops at the first gate (in circuit order) of a parameter that is not frozen
  std::vector<char> param_frozen;
  bool stop_at_first_live_gate = true;  // (engine.cpp plans both ways and keeps the cheaper: stopping early forgoes the pruning)
  bool frozen(int param_idx) const { return param_idx >= 0 && size_t(param_idx) < param_frozen.size() && param_frozen[size_t(param_idx)]; }
  int n_ops = 0;
  std::vector<PauliTerm> terms;
};

// Tile geometry for a given tile size.

// Builds the forward plan (circuit passes + measurement) or, with adjoint =
// true, the backward plan over (psi, lambda) tile pairs.  `tile_bits` = 0
// selects automatically.  Returns false and fills `err` on failure.
// `meas_tile_bits`: tile size of measurement-only passes (0 = the largest the forward kernel has).
bool build_plan(const Model& m, int tile_bits, int round_bits, bool adjoint, Plan* out,
                std::string* err, int full_threshold = 60, int meas_tile_bits = 0, bool cph_wave_bits = true,
                bool relabel = false, int wide_last_pass = -1, const std::vector<uint32_t>* forced_order = nullptr);

std::string describe_plan(const Plan& p);

}  // namespace qhbm
