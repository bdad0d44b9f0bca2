// program.h -- encoding shared by the host scheduler and the gfx950 kernels.
//
// A forward (or adjoint) evaluation is a short list of PASSES.  One pass is
// one kernel launch in which every workgroup owns one TILE of one statevector:
// 2^K amplitudes selected by K "local" bit positions of the amplitude index,
// resident in LDS for the whole pass.  The pass executes a PROGRAM (a few
// hundred 32-bit words, wave-uniform) on its tile:
//
//   OP_ROUND    every thread loads 2^R amplitudes (R "register bits" out of the
//               K local bits) and applies a sequence of INSTANCES in registers,
//               then writes back.  An instance is a fixed template of slots:
//                 MAT  x R        one-qubit gate on register bit j (X**t, Y**t, dense 2x2)
//                 PH1  x R        phase on amplitudes whose register bit j is 1 (Z**t)
//                 PH2  x R(R-1)/2 phase where register bits j and j' are both 1 (CZ**t)
//                 CPH  x 2R       phase where register bit j is 1 AND a thread bit / a
//                                 tile (non-local) bit is 1 (CZ**t across the boundary)
//               Diagonal gates therefore never cost an LDS sweep or a table:
//               they ride along with the round that already holds one of their
//               bits in registers.
//   OP_GATE2    dense two-qubit gate applied directly on the LDS tile.
//   OP_MEASURE  Pauli-sum expectation contributions of this tile.
//
// Internal amplitude-index convention: qubit q  <->  bit (n-1-q), so an index
// read as a binary number is the bitstring (cirq big-endian).
#pragma once
#include <stdint.h>

namespace qhbm {

constexpr int kMinTileBits = 10;  // states with fewer qubits are padded with idle qubits
constexpr int kMaxTileBits = 14;
constexpr int kMaxQubits = 32;    // amplitude indices are 32-bit
constexpr int kMaxOps = 1024;           // observables per engine (LDS accumulators)
constexpr int kMaxSlotsPerPass = 2048;  // gradient slots one adjoint pass may own

// ---- pass flags ------------------------------------------------------------
enum : uint32_t {
  PASS_INIT_BASIS = 1u << 0,  // tile := |bits> instead of loading it
  PASS_STORE = 1u << 1,       // write the tile back at the end
  PASS_ADJOINT = 1u << 2,     // tile pair (psi, lambda), program is a backward program
};

// ---- opcodes (low 8 bits of an instruction's first word) --------------------
enum : uint32_t {
  OP_END = 0,
  // [op | n_inst<<8] [regmask] [first_record]: n_inst consecutive fixed-layout RECORDS in the
  // coefficient buffer (see RecordLayout) drive the round.
  OP_ROUND = 1,
  OP_MEASURE = 3,  // [op | n_groups<<8] then groups x {[xl] [n_terms] terms x {[zl] [zn] [coef bits] [op_idx | ny<<24]}}
  OP_GATE2 = 4,    // [op | kind<<8] [pos_q0 | pos_q1<<8 (local bits)] [coef_off] [slot]
};
constexpr int kGate2Words = 4;
constexpr int kMeasTermWords = 4;

// ---- one-qubit micro-ops / coefficient-job tags ------------------------------
enum : uint32_t {
  MOP_X = 1,      // c*I - i*s*X on a register bit     (coef: c, s)
  MOP_Y = 2,      // c*I - i*s*Y                         (coef: c, s)
  MOP_MAT1 = 3,   // dense 2x2                           (coef: 8 floats, row-major re,im)
  MOP_MAT2 = 4,   // coefficient-job tag of a dense 4x4 (32 floats); executed by OP_GATE2
  MOP_PHASE = 5,  // coefficient-job tag of a diagonal term: (cos, sin) of pi * mult * t
};

constexpr int pair_index(int lo, int hi) { return hi * (hi - 1) / 2 + lo; }  // lo < hi

// One INSTANCE = one fixed-layout record of 32-bit words in the coefficient buffer, so that a
// wave fetches it with one coalesced load per 64 words (lane i <- word i, prefetched one
// instance ahead) and picks fields with v_readlane at compile-time lane indices -- no
// per-entry pointer chasing.  Static words (masks, predicates, gradient slots) are written once
// at upload; the coefficient words are rewritten by prep_coefs_kernel on every call.
//   word 0   x_mask | ph1_mask<<8 | ph2_mask<<16          word 1   cph_mask | y_mask<<16 | dense_mask<<24
//   X[R]{c,s}  PH1[R]{c,s}  PH2[NP]{c,s}  CPH[2R]{c,s}  CPHPRED[2R]  Y[R]{c,s}      (NP = R(R-1)/2)
//   CPHPRED = pos | kind<<8; kind 0 = local thread bit, kind 1 = tile (non-local) index bit
// then, 64-word aligned: dense 2x2 blocks DENSE[R] (8 floats; adjoint: U^dagger then generator,
// 16 floats) and, adjoint only, SLOT[6R+NP] = gradient slot per entry in the order
// X, Y, DENSE, PH1, PH2, CPH.
struct RecordLayout {
  int R, NP;
  bool adjoint;
  constexpr RecordLayout(int r, bool adj) : R(r), NP(r * (r - 1) / 2), adjoint(adj) {}
  constexpr int x(int j) const { return 2 + 2 * j; }
  constexpr int ph1(int j) const { return 2 + 2 * R + 2 * j; }
  constexpr int ph2(int pi) const { return 2 + 4 * R + 2 * pi; }
  constexpr int cph(int k) const { return 2 + 4 * R + 2 * NP + 2 * k; }
  constexpr int pred(int k) const { return 2 + 8 * R + 2 * NP + k; }
  constexpr int y(int j) const { return 2 + 10 * R + 2 * NP + 2 * j; }
  constexpr int base_words() const { return 2 + 12 * R + 2 * NP; }              // 62 (R=4), 82 (R=5)
  constexpr int base_vecs() const { return (base_words() + 63) / 64; }
  constexpr int dense_words() const { return adjoint ? 16 : 8; }
  constexpr int dense(int j) const { return 64 * base_vecs() + dense_words() * j; }
  constexpr int dense_vecs() const { return (R * dense_words() + 63) / 64; }
  constexpr int slot0() const { return 64 * (base_vecs() + dense_vecs()); }
  // slot order: X[R] Y[R] DENSE[R] PH1[R] PH2[NP] CPH[2R]
  constexpr int slot_x(int j) const { return slot0() + j; }
  constexpr int slot_y(int j) const { return slot0() + R + j; }
  constexpr int slot_dense(int j) const { return slot0() + 2 * R + j; }
  constexpr int slot_ph1(int j) const { return slot0() + 3 * R + j; }
  constexpr int slot_ph2(int pi) const { return slot0() + 4 * R + pi; }
  constexpr int slot_cph(int k) const { return slot0() + 4 * R + NP + k; }
  constexpr int vecs() const { return base_vecs() + dense_vecs() + (adjoint ? 1 : 0); }
  constexpr int words() const { return 64 * vecs(); }
};

// Lowered operation kinds produced by the host.
enum LoweredType : int { LOW_SKIP = 0, LOW_DIAG = 1, LOW_MAT1 = 2, LOW_MAT2 = 3 };

// Kernel argument block of one pass.
struct PassArgs {
  uint32_t flags;
  uint32_t n;              // qubits after padding
  uint32_t c;              // low `c` local bits are index bits 0..c-1 (contiguous in HBM)
  uint32_t n_nonlocal;     // n - K
  uint32_t prog_off;       // word offset of this pass's program
  uint32_t spread_off;     // offset of spread_hi[2^(K-c)] in the tables buffer
  uint32_t n_ops;          // observables (row length of out)
  uint32_t slot_base;      // adjoint: first gradient slot of this pass
  uint32_t n_slots;        // adjoint: gradient slots written by this pass
  uint8_t nonlocal_pos[32];  // ascending bit positions of the nonlocal index bits
  uint8_t local_pos[16];     // ascending bit positions of the K local index bits
};

}  // namespace qhbm
