// program.h -- encoding shared by the host scheduler and the gfx950 kernels.
//
// A forward (or adjoint) evaluation is a short list of PASSES.  One pass is
// one kernel launch in which every workgroup owns one TILE of one statevector:
// 2^K amplitudes selected by K "local" bit positions of the amplitude index,
// resident in LDS for the whole pass.  The pass executes a PROGRAM (a few
// hundred 32-bit words, wave-uniform) on its tile:
//
//   OP_ROUND    load 2^R amplitudes per thread (R "register bits" out of the K
//               local bits), apply a list of one-/two-qubit micro-ops in
//               registers, write back.
//   OP_DIAG     multiply by a product of diagonal gates; the phase of a
//               local index is factored into two small LDS tables built per
//               tile (low 7 local bits / remaining local bits) plus explicit
//               cross terms.
//   OP_MEASURE  Pauli-sum expectation contributions of this tile.
//
// Internal amplitude-index convention: qubit q  <->  bit (n-1-q), so an index
// read as a binary number is the bitstring (cirq big-endian).
#pragma once
#include <stdint.h>

namespace qhbm {

constexpr int kLoBits = 7;        // local bits covered by the E_lo phase table
constexpr int kMinTileBits = 10;  // states with fewer qubits are padded with idle qubits
constexpr int kMaxTileBits = 14;
constexpr int kMaxQubits = 32;    // amplitude indices are 32-bit
constexpr int kMaxCrossTerms = 64;      // cross (lo x hi) terms per OP_DIAG
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
  OP_ROUND = 1,    // [op | n_micro<<8] [regmask] then n_micro x {[mop|rb0<<8|kind<<16] [coef_off] [slot]}
  OP_DIAG = 2,     // [op] [n_lo | n_hi<<10 | n_cross<<20] then terms x {[lmask|par<<31] [nmask] [angle_idx] [slot]}
  OP_MEASURE = 3,
  OP_GATE2 = 4,    // [op | kind<<8] [pos_q0 | pos_q1<<8 (local bits)] [coef_off] [slot]  // [op | n_groups<<8] then groups x {[xl] [n_terms] terms x {[zl] [zn] [coef bits] [op_idx | ny<<24]}}
};
constexpr int kMicroWords = 3;
constexpr int kGate2Words = 4;
constexpr int kDiagTermWords = 4;
constexpr int kMeasTermWords = 4;

// ---- micro-ops inside a round ------------------------------------------------
enum : uint32_t {
  MOP_X = 1,     // c*I - i*s*X on register bit rb0   (coef: c, s)
  MOP_Y = 2,     // c*I - i*s*Y                         (coef: c, s)
  MOP_MAT1 = 3,  // general 2x2 on rb0                  (coef: 8 floats, row-major re,im)
  MOP_MAT2 = 4,  // coefficient-job tag of a dense 4x4 (32 floats); executed by OP_GATE2 on LDS
};

// Lowered operation kinds produced by the host (one per circuit gate).
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
