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
  // [op | n_inst<<8] [regmask] then n_inst instances:
  //   [x_mask | ph1_mask<<8 | ph2_mask<<16] [cph_mask | y_mask<<16 | dense_mask<<24]
  //   then one entry per set bit, in execution order (forward: X, Y, DENSE, PH1, PH2, CPH;
  //   adjoint: CPH, PH2, PH1, X, Y, DENSE -- the host lays entries out in that order):
  //     X/Y/DENSE/PH1/PH2: [coef_off] [slot]
  //     CPH: [pred] [coef_off] [slot]     pred = pos | kind<<8; kind 0 = local thread bit,
  //                                       kind 1 = tile (non-local) index bit
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
