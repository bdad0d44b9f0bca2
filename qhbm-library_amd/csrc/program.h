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
constexpr int kMaxOps = 1024;           // observables per engine (LDS accumulators, 64-bit fixed point)
// Gradient slots one adjoint pass may own: one LDS cell per wave and slot (K = 12, 4 waves: 6 KiB next
// to the 32 KiB exchange tile, so four workgroups still share a CU's 160 KiB).
constexpr int kMaxSlotsPerPass = 384;
// Expectation values are accumulated across waves, tiles and passes as 64-bit FIXED-POINT integers
// (integer addition is associative: the result does not depend on the order in which workgroups
// finish, so a value is bit-identical from run to run and for any sharding of the batch).  An op's
// partial sums are bounded by B = sum_k |c_k|; they are scaled by 2^(kValueFracBits - ceil(log2 B)).
constexpr int kValueFracBits = 40;

// ---- pass flags ------------------------------------------------------------
enum : uint32_t {
  PASS_INIT_BASIS = 1u << 0,  // tile := |bits> instead of loading it
  PASS_STORE = 1u << 1,       // write the tile back at the end
  PASS_ADJOINT = 1u << 2,     // tile pair (psi, lambda), program is a backward program
  PASS_GENERAL = 1u << 3,     // the program uses Y / dense 2x2 / dense two-qubit ops (rare-path kernel variant)
  PASS_SKIP_MEASURE = 1u << 4,  // forward: ignore the measurement groups (the values come from lambda = O psi)
  PASS_RELABEL = 1u << 5,       // adjoint: the store moves the index bits this pass finished (PassArgs::relabel_*)
  PASS_NO_ZERO_FILL = 1u << 6,  // forward, with PASS_INIT_BASIS: tiles without the basis amplitude write nothing (later
                                // passes clear what they load of them: PassArgs::frozen_old_local)
};

// ---- opcodes (low 8 bits of an instruction's first word) --------------------
enum : uint32_t {
  OP_END = 0,
  // [op | n_inst<<8] [regmask] [first_record] [tl_table]: n_inst consecutive fixed-layout RECORDS in
  // the coefficient buffer (see RecordLayout) drive the round; tl_table = offset (in the pass's part
  // of the tables buffer, PassArgs::tl_off) of the round's thread -> local-index table TL[tid]: the
  // bits of tid deposited on the non-register local bits, computed once by the scheduler instead of
  // ~50 VALU per thread and round.  Word 4 (adjoint): local index bits that (a) spell the WAVE index in
  // this round and (b) have no non-diagonal gate left to un-apply -- psi is zero wherever such a bit
  // differs from the input bitstring, so a wave whose bits differ skips the round's instances.
  OP_ROUND = 1,
  OP_MEASURE = 3,  // [op | n_groups<<8] then groups x {[xl] [n_terms] terms x {[zl] [zn] [coef bits] [op_idx | ny<<24]}}
  OP_GATE2 = 4,    // [op | kind<<8] [pos_q0 | pos_q1<<8 (local bits)] [coef_off] [slot]
  // Many DIAGONAL terms (x = 0: Z strings -- the shards of a modular Hamiltonian, energy.py:165-167 / 200-209) at
  // once: the tile's |psi|^2 goes through a Walsh-Hadamard transform over the register and lane bits of its local
  // index, after which term k is ONE coefficient, W[z_k], times the parity of its wave and tile bits -- a few hundred
  // instructions per wave for any number of terms instead of ~30 per term.
  // [op | n_terms<<8] [class_end x 16] then terms x {[zl] [zn] [coef bits] [op_idx]} sorted by class = zl >> (K - R),
  // the register part of the mask (class_end[c] = end of class c in the term list).
  OP_MEASURE_WHT = 5,
};
constexpr int kWhtHeaderWords = 1 + 16;  // (kRoundBits = 4 register bits: 16 classes)
constexpr uint32_t kWhtMinTerms = 32;  // below: the per-term path (a ZZ chain of one operator is cheaper there)
constexpr uint32_t kRoundNoBarrier = 1u << 31;  // OP_ROUND word 0: the next op is a round whose waves own the same amplitudes
constexpr int kRoundWords = 5;  // [op | n_instances<<8 | flags] [register-bit mask] [first record] [tl_table] [dead mask]
constexpr int kGate2Words = 4;
constexpr int kMeasTermWords = 4;

// ---- one-qubit micro-ops / coefficient-job tags ------------------------------
enum : uint32_t {
  MOP_X = 1,      // c*I - i*s*X on a register bit     (coef: tan(theta/2), sin(theta): three shears)
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
// at upload; coefficient words are rewritten on every call by prep_coefs_kernel (one (cos, sin)
// or matrix per gate) and combine_diag_kernel (the FULL table).  Rounds hold R = 4 register bits.
//   vec 0:  [0] x_mask | fph1<<4 | ph1_mask<<8 | x_slot_mask<<12 (adjoint: X gates that own a gradient slot) | ph2_mask<<16 | fph2<<24
//           [1] cph_mask | y_mask<<16 | dense_mask<<24 | FULL<<31
//           X[4]{c,s}  CPH[8]{c,s}  CPHPRED[8]  then EITHER  PH1[4]{c,s} PH2[6]{c,s} Y[4]{c,s}
//                                               OR (FULL)  FULL[15]{c,s}
//           CPHPRED = pos | kind<<8: bit `pos` of the index word TL[tid] | tile_id << K -- kind 0 = a local thread
//           bit (pos < K), kind 1 = a tile (non-local) index bit (pos = K + its rank among the non-local bits); the
//           kernels read `pos` only, `kind` serves the host's accounting
//           FULL[m-1] = product of the instance's PH1/PH2 phases whose bits are all set in the
//           register value m: the whole diagonal on the register bits is then ONE complex
//           multiply per amplitude.  A FULL instance keeps its term masks in fph1/fph2 and has
//           ph1_mask = ph2_mask = 0, so table and per-term slots are independent triangles.
//   vec 1:  (FULL only) the per-term phases PH1[4] PH2[6] that combine_diag_kernel multiplies
//   vec 2:  DENSE[4] 2x2 blocks (8 floats; adjoint: U^dagger then generator, 16 floats)
//   vec 3:  (adjoint) SLOT[32] = gradient slot per entry, pass-local, laid out for the eight-wide wave
//           reduction (slot_lane8): group 0 = X + PH1, 1 = PH2, 2 = CPH, 3 = Y + DENSE
// Everything a round touches per instance sits in vec 0 (one coalesced 256-byte load per wave).
constexpr int kRoundBits = 4;
struct RecordLayout {
  int R, NP;
  bool adjoint;
  constexpr RecordLayout(int r, bool adj) : R(r), NP(r * (r - 1) / 2), adjoint(adj) {}
  constexpr int x(int j) const { return 2 + 2 * j; }
  constexpr int cph(int k) const { return 10 + 2 * k; }
  constexpr int pred(int k) const { return 26 + k; }
  constexpr int full(int m) const { return 34 + 2 * (m - 1); }  // m = 1..15
  constexpr int ph1(int j) const { return 34 + 2 * j; }
  constexpr int ph2(int pi) const { return 42 + 2 * pi; }
  constexpr int y(int j) const { return 54 + 2 * j; }
  constexpr int in_ph1(int j) const { return 64 + 2 * j; }    // FULL: combine inputs
  constexpr int in_ph2(int pi) const { return 72 + 2 * pi; }
  constexpr int dense_words() const { return adjoint ? 16 : 8; }
  constexpr int dense(int j) const { return 128 + dense_words() * j; }
  constexpr int slot0() const { return 192; }
  // Gradient slots of an instance are reduced over the wave EIGHT at a time (kernels.hip add_slots8):
  // value v of slot group g8 ends up in the lanes with bits (2, 3, 4) = v and bits (0, 1, 5) = g8, and
  // the slot-vector word of that lane IS the slot.  Groups: 0 = X[4] + PH1[4], 1 = PH2[6],
  // 2 = CPH[8], 3 = Y[4] + DENSE[4].
  constexpr int slot_lane8(int g8, int v) const {
    return slot0() + ((g8 & 3) | ((v & 3) << 2) | ((v >> 2) << 4) | ((g8 >> 2) << 5));
  }
  constexpr int slot_x(int j) const { return slot_lane8(0, j); }
  constexpr int slot_ph1(int j) const { return slot_lane8(0, 4 + j); }
  constexpr int slot_ph2(int pi) const { return slot_lane8(1, pi); }
  constexpr int slot_cph(int k) const { return slot_lane8(2, k); }
  constexpr int slot_y(int j) const { return slot_lane8(3, j); }
  constexpr int slot_dense(int j) const { return slot_lane8(3, 4 + j); }
  // 3 / 5 vectors: an odd count keeps the 256-byte record vectors that every wave of the chip
  // streams at the same time spread over all L2 channels (a 1 KiB stride hits every 4th).
  constexpr int vecs() const { return adjoint ? 5 : 3; }
  constexpr int words() const { return 64 * vecs(); }
};
constexpr uint32_t kFullDiagFlag = 1u << 31;

// Lowered operation kinds produced by the host.
enum LoweredType : int { LOW_SKIP = 0, LOW_DIAG = 1, LOW_MAT1 = 2, LOW_MAT2 = 3 };

// Kernel argument block of one pass.
struct PassArgs {
  uint32_t flags;
  uint32_t n;              // qubits after padding
  uint32_t c;              // low `c` local bits are index bits 0..c-1 (contiguous in HBM)
  uint32_t n_nonlocal;     // n - K
  // Tiles that differ from the input bitstring on a bit of `zero_mask` are identically zero and are NOT
  // LAUNCHED: the grid holds 2^n_free tiles per state, the other n_nonlocal - n_free tile-id bits are read
  // off the input bitstring (kernels.hip launched_tile).
  uint32_t n_free;
  uint32_t prog_off;       // word offset of this pass's program
  uint32_t spread_off;     // offset of spread_hi[2^(K-c)] in the tables buffer
  uint32_t tl_off;         // offset of this pass's round TL tables in the tables buffer (OP_ROUND word 3)
  uint32_t zero_mask;      // adjoint: index bits non-local in this and every later pass (see fill_args)
  uint32_t spread_shift;   // local bits above c contiguous from bit s: spread_hi[j] = j << s (no lookup); else ~0u
  uint32_t n_ops;          // observables (row length of out)
  uint32_t slot_base;      // adjoint: first gradient slot of this pass
  uint32_t n_slots;        // adjoint: gradient slots written by this pass (row length of tile_grad)
  // Batched programs (parameter-shift): batch element e = blockIdx.x >> n_nonlocal runs program
  // e / prog_states on state (state0 + e % prog_states), reads its coefficients at
  // coef + (e / prog_states) * coef_stride and writes output row e.  0 = one program, rows by state.
  uint32_t prog_states;
  uint32_t coef_stride;    // floats between the coefficient buffers of consecutive programs
  uint8_t nonlocal_pos[32];  // ascending PHYSICAL bit positions of the nonlocal index bits
  uint8_t local_pos[16];     // ascending PHYSICAL bit positions of the K local index bits
  // Adjoint plans that relabel (schedule.h Pass): where logical index bit b (bit n-1-q <-> qubit q, the bit
  // of the input bitstring) sits in the addresses this pass loads; identity otherwise.
  uint8_t phys_of[32];
  // local index bits whose != input half holds STALE data, zeroed at load: adjoint -- bits finished and moved by
  // earlier passes; forward -- bits no earlier pass has acted on (never written: PASS_NO_ZERO_FILL)
  uint32_t frozen_old_local;
  // PASS_RELABEL: the store writes only the amplitudes whose bits `frozen_new_local` equal the input
  // bitstring, live out-local index o -> tables[relabel_off + 2 o] = local index (those bits clear),
  // [.. + 1] = offset in the state; the finished bits land on `fz_out_pos` carrying the input bit.
  uint32_t frozen_new_local;
  uint32_t relabel_off;
  uint32_t relabel_pairs;     // 1: consecutive o are adjacent in HBM (16-byte stores), 0: 8-byte stores
  uint32_t n_fz;
  uint8_t fz_local_bit[16];
  uint8_t fz_out_pos[16];
  // The per-bit tables above in the forms a workgroup can use WITHOUT walking them (kernels.hip input_index,
  // launched_tile, tile_base_of, pass_tile_ctx: one table element per lane and a ballot instead of a loop of dependent byte
  // loads): the inverse of phys_of (log_of[p] = the logical bit on physical position p), the nonlocal positions
  // as a mask, and the index offsets of a thread's eight float4 rows, tile_offset(I << (K - 3)) for I = 0..7.
  uint8_t log_of[32];
  uint32_t nonlocal_mask;
  uint32_t row_off[8];
  // PASS_RELABEL, the same way: the relabel table is OR-decomposable in the out-index o, so a thread looks up ONE
  // entry (its own low part of o) and adds the entries of the high parts from here: (local index, offset) of
  // o = i << relabel_hi_shift for the relabel_iters iterations of the store; the local index of o = 1 (the upper
  // amplitude of a 16-byte pair); fz_src[p] = the local bit whose input value lands on address position p (0xff: none).
  uint32_t relabel_hi[16][2];
  uint32_t relabel_iters;
  uint32_t relabel_l1;
  uint8_t fz_src[32];
};

}  // namespace qhbm
