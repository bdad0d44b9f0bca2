// observable.hip -- lambda = O psi and <psi|O|psi> for Pauli sums with MANY X-masks (gfx950).
//
// The reference hands every PauliSum to tfq.layers.Expectation (/root/reference/qhbmlib/inference/qnn.py:120-139);
// the adjoint backward starts from lambda = sum_k upstream[s, op_k] c_k P_k psi, and with the state in HBM anyway
// the same sweep yields the values <psi|O_t|psi>.
//
//   (P psi)[j] = i^ny (-1)^{popc((j ^ x) & z)} psi[j ^ x]
//
// apply_observable_kernel (kernels.hip) gathers psi[j ^ x] from L2 once per distinct mask x and block of 2048
// amplitudes: fine for a chain Hamiltonian (20 masks), 453 gathers per block for BASELINE config 4 (512 random
// strings on 24 qubits, 480 masks) -- 61 GB through L2 per 128-MiB state.  Here a workgroup owns a BLOCK of 2^13
// amplitudes, a mask is split x = x_out | x_in (bits outside / inside the block), terms arrive sorted by x_out and
// cut into GROUPS of equal x_out: the partner block b ^ x_out of a group is fetched ONCE (coalesced 16-byte buffer
// loads, staged through registers) into LDS, and every mask of the group reads it from there at l ^ x_in with
// ds_read_b128 (256 B/clk/CU against the 64 B/clk/CU a CU gets from L2).  Config 4: 173 block fetches instead of 453
// gathers.
//
// One workgroup of 1024 threads per CU, as two HALVES of 512.  Thread t of either half owns the same 8 ADJACENT PAIRS
// of the block, pair index t + 512 p (p = 0..7), i.e. local amplitude l = 2 t + 1024 p + h: slot bits {0, 10, 11,
// 12}, thread bits 1..9 -- and the two halves share every fetched block but split its MASKS (the scheduler deals a
// group's masks to the halves by term count), each into its own accumulators, merged through LDS at the end.
// Why one fat workgroup instead of two of 512 threads per CU (the first version): the blocks an XCD has in flight --
// one per workgroup -- read, group by group, the same "window" of partner blocks (b ^ x_out for neighbouring b), and
// that window must survive in the 4-MiB L2 from one group to the next while the workgroups drift apart.  64 blocks of
// 64 KiB in flight per XCD make a 4-MiB window: measured hit rate 0.61 (0.18 with eight XCDs on one state), fabric
// reads 9.4 GB per 128-MiB state, the kernel bound by them.  32 blocks in flight: 0.76, 5.5 GB -- what the count of
// distinct windows predicts.  (Two workgroups SHARING a block and taking alternate groups drift apart by whole windows
// -- 0.13; a resident grid walking the blocks with a stride loses the compact sliding window -- both measured, both
// dropped.)  The halves of one workgroup are kept in step by its barrier.
// (Round 6, BB = 12 below: blocks of 2^12 under two INDEPENDENT workgroups of 512 threads per CU -- the same 2-MiB footprint
// per XCD, one workgroup's store burst under the other's masks -- measured 94.4 against 51.2 ms: co-resident workgroups
// drift apart by whole windows, hit rate 0.27, 20 GB of fabric reads.  profiles/r06_obs_block_bits.txt; option-only.)
// (Round 6, BB = kObsShapeRows: this workgroup, the halves splitting the block's ROWS instead of a group's masks -- every mask
// on both halves, four pairs per thread, no idle half, no merge, 72 - 96 registers: 57.6 against 51.3 ms; with the spare
// registers holding the partner rows of a second mask, two masks per LDS wait: 67.2.  The mask phase is bound by instruction
// issue, not by a wave's LDS latency: what halves the work per instruction or adds branches loses.  Option-only.)
// Pipeline per group: the block of group s + 1 goes from registers to the OTHER of two 64-KiB LDS buffers while the
// masks of group s are applied from the first; the blocks of groups s + 2 and s + 3 are in flight in the two
// four-row prefetch sets of every thread (128 KiB per CU); ONE barrier per group.  The two halves do the store burst
// and their masks in opposite order, so that each half's 32 KiB of ds_write_b128 run under the other half's masks.
// The partner of an adjacent pair is an adjacent pair, so every LDS read is one 16-byte word, conflict-free (an XOR
// on the lane bits permutes the 16-byte chunks of a lane group).  x_in's slot part selects WHICH row (p ^ xp) and
// which half (h ^ x_0) a slot pairs with: rows through an 8-way scalar switch over ds_read offsets, halves at
// compile time.
// Sign of a term at own index j: (-1)^{popc(j & z) + ny}; the block part is scalar, the thread part one popcount
// per term and thread, and the slot part -- parity(a & zs) for slot a, zs = the four slot bits of z -- is COMPILE
// TIME: the accumulation of a term is one of 64 straight-line variants (zs x odd x imaginary), 16 packed FMAs
// whose +-w / re<->im choices are op_sel modifiers on a (w, -w) register pair.  No per-amplitude sign arithmetic.
//
// Values.  <psi|P|psi> is real and the (j, j ^ x) and (j ^ x, j) contributions are complex conjugates, so the
// value-only modes run a group with x_out != 0 on ONE block of each pair only, weight 2 (half the fetches): the block
// whose pivot bit of x_out is clear, the pivot taken below the bits that select the XCD (obs_next_group).
// One partial per (state, block, op) leaves in value_part; value_parts_blocks_kernel adds a state's partials in
// block order in fp64 into the fixed-point accumulators (bit-reproducible, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "device_common.h"
#include "kernels.h"

namespace qhbm {

namespace {

constexpr uint32_t kOH = 512;                                // threads of one half: one per column of adjacent pairs
// The two shapes (kernels.h): BB = 13 -- one workgroup of two halves per CU, eight pairs per thread; BB = 12 -- two
// independent workgroups of 512 threads per CU, four pairs per thread, every mask of a group applied by the one "half".
// BB = kObsShapeRows: the block of 2^13 and the one workgroup of the first shape, but the halves split the block's ROWS (half hh
// owns rows 4 hh .. 4 hh + 3 of every pair column: four pairs per thread) and each applies EVERY mask of a group to its
// rows -- no idle half in a group of one mask, no merge at the end, and half the accumulator and partner registers.
template <int BB>
struct ObsShape {
  static_assert(BB == kObsBlockBits || BB == kObsBlockBitsSmall || BB == kObsShapeRows, "three shapes");
  static constexpr int kBits = BB == kObsBlockBitsSmall ? kObsBlockBitsSmall : kObsBlockBits;
  static constexpr bool kHalves = BB != kObsBlockBitsSmall;       // a workgroup of two halves
  static constexpr bool kByTerms = BB == kObsBlockBits;           // the halves split a group's MASKS (else: the rows, or nothing)
  static constexpr bool kByRows = BB == kObsShapeRows;
  static constexpr uint32_t kThreads = kHalves ? 2 * kOH : kOH;   // per workgroup
  static constexpr uint32_t kPairs = kByTerms ? 8 : 4;            // adjacent pairs per thread
  static constexpr uint32_t kBlock = 1u << kBits;                 // amplitudes per block
  static constexpr uint32_t kWaves = kThreads / 64;
  static constexpr uint32_t kBuf = kBlock / 2;                    // 16-byte words of one LDS block buffer
  static_assert(kOH * kPairs * 2 * (kByRows ? 2 : 1) == kBlock, "512 threads x their pairs (x 2 halves of rows) = one block");
};

// 16-byte words as a NATIVE vector type: copies of HIP's v4f struct become memcpy calls between address spaces,
// which keep the register arrays below in scratch memory
typedef float v4f __attribute__((ext_vector_type(4)));


// ---- one term on eight slots: a jump into a table of 64 straight-line variants ---------------------
// (scripts/gen_observable_asm.py writes observable_variants.inc and documents the layout.)  A C++ switch over the
// variants makes the structuriser copy the whole accumulator file into temporaries around every case -- 32 moves
// for 16 FMAs, and the register file overflows; inside one asm statement the accumulators are updated in place.
// w = (w, -w): op_sel on source 0 picks the sign per result half, op_sel on source 1 swaps re / im for an
// imaginary weight.  `off` = 68 * variant + 12 (wave-uniform).  vcc carries the jump target.
#include "observable_variants.inc"
static_assert(OBS_CHUNK_BYTES == kObsChunkBytes && OBS_PREAMBLE_BYTES == kObsPreambleBytes, "engine.cpp builds the jump offsets");
__device__ __forceinline__ void obs_fma8(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f& a4, v2f& a5, v2f& a6, v2f& a7, v2f v0,
                                         v2f v1, v2f v2, v2f v3, v2f v4, v2f v5, v2f v6, v2f v7, v2f w, uint32_t off) {
  asm(OBS_ASM_FMA8
      : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7)
      : [w] "v"(w), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [v4] "v"(v4), [v5] "v"(v5), [v6] "v"(v6),
        [v7] "v"(v7), [off] "s"(off)
      : "vcc", "scc");
}
// value modes: s_{i & 3} += (+-)(own_i . v_i) (imaginary weight: own.y v.x - own.x v.y); the two halves of every s
// are added, and the sum weighted, after the term
__device__ __forceinline__ void obs_dot8(v2f& s0, v2f& s1, v2f& s2, v2f& s3, v2f o0, v2f o1, v2f o2, v2f o3, v2f o4, v2f o5,
                                         v2f o6, v2f o7, v2f v0, v2f v1, v2f v2, v2f v3, v2f v4, v2f v5, v2f v6, v2f v7,
                                         uint32_t off) {
  asm(OBS_ASM_DOT8
      : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3)
      : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [o4] "v"(o4), [o5] "v"(o5), [o6] "v"(o6), [o7] "v"(o7),
        [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [v4] "v"(v4), [v5] "v"(v5), [v6] "v"(v6), [v7] "v"(v7),
        [off] "s"(off)
      : "vcc", "scc");
}

__device__ __forceinline__ v2f lo_half(const v4f& r) { return v2f{r.x, r.y}; }
__device__ __forceinline__ v2f hi_half(const v4f& r) { return v2f{r.z, r.w}; }

// One term on the thread's 16 slots (slot a = 2 p + h: row p, half h).  a[]: lambda accumulators (ACC) or the thread's
// own amplitudes (value modes); r[p]: the partner pair of row p as read from LDS.  off0 / off1: table offsets of the
// variant for slots 0..7 / 8..15 (they differ in the base sign when the Z mask holds the highest slot bit).
template <bool ACC>
__device__ __forceinline__ void obs_term(v2f (&a)[16], const v4f (&r)[8], v2f w, v2f (&s)[4], uint32_t off0, uint32_t off1) {
  if constexpr (ACC) {
    obs_fma8(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], lo_half(r[0]), hi_half(r[0]), lo_half(r[1]), hi_half(r[1]),
             lo_half(r[2]), hi_half(r[2]), lo_half(r[3]), hi_half(r[3]), w, off0);
    obs_fma8(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], lo_half(r[4]), hi_half(r[4]), lo_half(r[5]),
             hi_half(r[5]), lo_half(r[6]), hi_half(r[6]), lo_half(r[7]), hi_half(r[7]), w, off1);
  } else {
    obs_dot8(s[0], s[1], s[2], s[3], a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], lo_half(r[0]), hi_half(r[0]),
             lo_half(r[1]), hi_half(r[1]), lo_half(r[2]), hi_half(r[2]), lo_half(r[3]), hi_half(r[3]), off0);
    obs_dot8(s[0], s[1], s[2], s[3], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], lo_half(r[4]), hi_half(r[4]),
             lo_half(r[5]), hi_half(r[5]), lo_half(r[6]), hi_half(r[6]), lo_half(r[7]), hi_half(r[7]), off1);
  }
}

// blocks of 2^12: eight slots, four rows
template <bool ACC>
__device__ __forceinline__ void obs_term(v2f (&a)[8], const v4f (&r)[4], v2f w, v2f (&s)[4], uint32_t off0, uint32_t) {
  if constexpr (ACC) {
    obs_fma8(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], lo_half(r[0]), hi_half(r[0]), lo_half(r[1]), hi_half(r[1]),
             lo_half(r[2]), hi_half(r[2]), lo_half(r[3]), hi_half(r[3]), w, off0);
  } else {
    obs_dot8(s[0], s[1], s[2], s[3], a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], lo_half(r[0]), hi_half(r[0]),
             lo_half(r[1]), hi_half(r[1]), lo_half(r[2]), hi_half(r[2]), lo_half(r[3]), hi_half(r[3]), off0);
  }
}

#ifdef QHBM_OBS_TIMING  // diagnostic build: cycles of one workgroup's waves per phase, printed (never shipped)
#define OBS_T(slot) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); tacc[slot] += now_ - tlast; tlast = now_; }
#else
#define OBS_T(slot)
#endif
#ifndef QHBM_OBS_SKEW
#define QHBM_OBS_SKEW 1
#endif
#ifndef QHBM_OBS_ROWS_PAIRED
#define QHBM_OBS_ROWS_PAIRED 0   // (rows shape, 1: the partner rows of two masks per LDS wait -- measured slower: 67.2 against 57.6 ms)
#endif
#ifndef QHBM_OBS_LOAD_MOD
#define QHBM_OBS_LOAD_MOD ""  // cache-policy bits of the partner-block loads (A/B builds: " nt", " sc1", ...)
#endif
// The eight partner rows of a mask: row p pairs with row p ^ xp and thread t with t ^ xt, i.e. the LDS byte address of
// slot row p is (t << 4 | p << 13 | buffer) ^ xrow with xrow = xt << 4 | xp << 13 from the term record: one XOR per row.
// (Measured and dropped: the rows as immediate ds_read offsets behind an 8-way jump table like the term variants --
// 15 vector instructions fewer per term, the LDS wait inside the asm statement -- 51.4 -> 52.5 ms on config 4: the mask
// phase is bound by the LDS reads themselves, 8 KiB per wave and term, not by VALU issue.)
// The reads go through ABSOLUTE LDS addresses (the workgroup's dynamic array starts at LDS byte 0: the kernel has no static
// LDS, checked once per workgroup): `lds + offset` on the array's symbol costs a v_add_u32 with a link-time 0 per row -- eight
// instructions per mask in a phase that is bound by instruction issue.
typedef const v4f __attribute__((address_space(3))) * ObsLdsWord;
template <int N, int... P>
__device__ __forceinline__ void obs_rows_(v4f (&r)[N], const char*, uint32_t base, std::integer_sequence<int, P...>) {
  ((r[P] = *reinterpret_cast<ObsLdsWord>(uintptr_t(base ^ uint32_t(P << 13)))), ...);
}
template <int N>
__device__ __forceinline__ void obs_rows(v4f (&r)[N], const char* lds, uint32_t base) {
  obs_rows_(r, lds, base, std::make_integer_sequence<int, N>{});
}
// (register arrays are only ever indexed by compile-time constants: integer_sequence folds, never loops)
// One block (64 KiB) into the thread's prefetch registers: row P at `blk` + 8192 P + 16 tid, as BUFFER loads -- the
// block is the buffer (a wave-uniform descriptor in four SGPRs), the row a scalar offset, and the eight loads share
// ONE 32-bit offset register; flat global loads need eight 64-bit address pairs, which the register file of
// accumulators + partner rows + prefetch rows has no room for (18 spilled registers).
typedef v4f ObsSet[4];   // one prefetch set: the thread's 16-byte word of four rows
typedef int v4i __attribute__((ext_vector_type(4)));
// A thread fetches and stages FOUR rows of its pair column: rows 4 hh .. 4 hh + 3 for half hh (wave-uniform) of the eight of
// a block of 2^13, all four of a block of 2^12.
// The loads are volatile asm with MANUAL s_waitcnt: two prefetch sets are in flight across the loop's back edge, and
// the compiler's own wait insertion, exact inside straight-line code, gives up at the loop header -- it waited for
// vmcnt(0), i.e. for the set issued one step ago as well, which halves the prefetch distance (measured).  The
// compiler does not know these registers are pending: nothing may touch a set between obs_fetch and obs_wait_older
// (the sets are written and read by unconditional straight-line code only -- no phi, hence no copy).
typedef int v4i __attribute__((ext_vector_type(4)));
template <int BB>
__device__ __forceinline__ void obs_fetch(v4f (&pf)[4], const float2* __restrict__ blk, uint32_t t16, uint32_t row0) {
  // 0x00020000: the raw-buffer word 3 of gfx90a / gfx942 / gfx950 (32-bit data format, no swizzle)
  // (descriptor words: base[31:0] | base[47:32], stride 0 | bytes of the block | word 3)
  const uint64_t addr = reinterpret_cast<uint64_t>(blk);
  const v4i rsi = v4i{int(uint32_t(addr)), int(uint32_t(addr >> 32) & 0xffffu), 8 << ObsShape<BB>::kBits, 0x00020000};
  const uint32_t o0 = 8192u * row0, o1 = o0 + 8192u, o2 = o0 + 16384u, o3 = o0 + 24576u;
  asm volatile("buffer_load_dwordx4 %0, %4, %5, %6 offen" QHBM_OBS_LOAD_MOD "\n\t"
               "buffer_load_dwordx4 %1, %4, %5, %7 offen" QHBM_OBS_LOAD_MOD "\n\t"
               "buffer_load_dwordx4 %2, %4, %5, %8 offen" QHBM_OBS_LOAD_MOD "\n\t"
               "buffer_load_dwordx4 %3, %4, %5, %9 offen" QHBM_OBS_LOAD_MOD
               : "=&v"(pf[0]), "=&v"(pf[1]), "=&v"(pf[2]), "=&v"(pf[3])
               : "v"(t16), "s"(rsi), "s"(o0), "s"(o1), "s"(o2), "s"(o3));
}
// Every obs_fetch but the `NEWER` most recent ones has landed (4 loads each).  The set that is about to be staged goes
// THROUGH the statement ("+v"): its consumers depend on the wait by data flow.  (No "memory" clobber on these asm
// statements: an asm that may write memory makes the compiler load the term records -- read-only kernel arguments --
// through the vector path instead of the scalar one.)
template <int NEWER>
__device__ __forceinline__ void obs_wait_older(v4f (&pf)[4]) {
  static_assert(NEWER == 0 || NEWER == 1, "two sets");
  if constexpr (NEWER == 1) asm volatile("s_waitcnt vmcnt(4)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]));
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]));
}
template <int... J>
__device__ __forceinline__ void obs_stage_(v4f* dst, const v4f (&pf)[4], std::integer_sequence<int, J...>) {
  ((dst[512 * J] = pf[J]), ...);
}
__device__ __forceinline__ void obs_stage(v4f* dst, const v4f (&pf)[4]) {   // dst = buffer + t + 512 row0
  obs_stage_(dst, pf, std::make_integer_sequence<int, 4>{});
}
// the second half hands its accumulators to the first through an LDS buffer
template <int... P>
__device__ __forceinline__ void obs_acc_out_(v4f* dst, const v2f (&a)[16], std::integer_sequence<int, P...>) {
  ((dst[512 * P] = v4f{a[2 * P].x, a[2 * P].y, a[2 * P + 1].x, a[2 * P + 1].y}), ...);
}
template <int... P>
__device__ __forceinline__ void obs_acc_in_(v2f (&a)[16], const v4f* src, std::integer_sequence<int, P...>) {
  (([&] { const v4f o = src[512 * P]; a[2 * P] += v2f{o.x, o.y}; a[2 * P + 1] += v2f{o.z, o.w}; }()), ...);
}
template <int N, int... P>
__device__ __forceinline__ void obs_own_(v2f (&a)[N], const v4f* __restrict__ src, std::integer_sequence<int, P...>) {
  (([&] { const v4f o = src[512 * P]; a[2 * P] = v2f{o.x, o.y}; a[2 * P + 1] = v2f{o.z, o.w}; }()), ...);
}
template <int N, int... P>
__device__ __forceinline__ void obs_store_(v4f* __restrict__ dst, const v2f (&a)[N], std::integer_sequence<int, P...>) {
  // (non-temporal: lambda is read next by another kernel -- it must not displace the partner blocks in L2)
  ((__builtin_nontemporal_store(v4f{a[2 * P].x, a[2 * P].y, a[2 * P + 1].x, a[2 * P + 1].y}, dst + 512 * P)), ...);
}
template <int N, int... P>
__device__ __forceinline__ float obs_energy_(const v4f* __restrict__ src, const v2f (&a)[N], std::integer_sequence<int, P...>) {
  float e = 0.f;
  (([&] {
     const v4f o = src[512 * P];
     e += (o.x * a[2 * P].x + o.y * a[2 * P].y) + (o.z * a[2 * P + 1].x + o.w * a[2 * P + 1].y);
   }()), ...);
  return e;
}
template <int N, int... I>
__device__ __forceinline__ void obs_zero_(v2f (&a)[N], std::integer_sequence<int, I...>) { ((a[I] = v2f{0.f, 0.f}), ...); }

// first group at or after g that this block runs: value modes skip a pair's upper block
template <bool HALVE>
__device__ __forceinline__ uint32_t obs_next_group(const ObsBGroup* __restrict__ groups, uint32_t n_groups, uint32_t g, uint32_t bx,
                                                   uint32_t pivot_mask) {
  // Value modes: <psi|P|psi> is real, so of the two blocks a mask pairs only ONE applies it (weight 2) -- the one with
  // the group's PIVOT bit clear.  The pivot is the highest bit of x_out inside `pivot_mask`, else its highest bit:
  // when every XCD owns an eighth of each state (the top three block bits), a pivot among those bits would leave the
  // first XCD all the work of that group and the last one none (config 4: 148 against 32 groups per block, the
  // kernel as slow as without halving); below them the halves alternate inside every XCD's range.
  if constexpr (HALVE) {
    while (g < n_groups) {
      const uint32_t xo = groups[g].xout, m = xo & pivot_mask;
      if (xo == 0u || !((bx >> (31 - __builtin_clz(m ? m : xo))) & 1u)) break;
      ++g;
    }
  }
  return g;
}

template <int MODE>
struct ObsCtx {
  const ObsBTerm* terms;
  const float* up;
  const char* lds;   // the workgroup's LDS (byte address 0 of the dynamic array)
  float* cells;
  uint32_t n_ops, bx, t, tid;
  uint32_t row_bytes;   // LDS byte offset of the thread's first row (rows split between the halves: 0 or 4 x 8192)
  bool second;          // the thread's slots are 8..15 of the block layout (variant offset off1)
};

// One term on the thread's slots (2 NP of them: NP rows of adjacent pairs), from its 32-byte record.
template <int MODE, int NP>
__device__ __forceinline__ void obs_term_rows(const ObsCtx<MODE>& c, const ObsBTerm& t, uint32_t cur_bytes, v4f (&r)[NP]) {
  obs_rows(r, c.lds, ((c.t << 4) | c.row_bytes | cur_bytes) ^ t.xrow);
}
template <int MODE, int NP>
__device__ __forceinline__ void obs_term_math(const ObsCtx<MODE>& c, const ObsBTerm t, float pair_weight,
                                              v2f (&a)[2 * NP], v4f (&r)[NP], v2f& d2, uint32_t& cur_op, v2f (&dq)[4]);
template <int MODE, int NP>
__device__ __forceinline__ void obs_one_term(const ObsCtx<MODE>& c, const ObsBTerm t, uint32_t cur_bytes, float pair_weight,
                                             v2f (&a)[2 * NP], v4f (&r)[NP], v2f& d2, uint32_t& cur_op, v2f (&dq)[4]) {
  if (t.meta & kObsNewMask) obs_term_rows<MODE, NP>(c, t, cur_bytes, r);
  obs_term_math<MODE, NP>(c, t, pair_weight, a, r, d2, cur_op, dq);
}
template <int MODE, int NP>
__device__ __forceinline__ void obs_term_math(const ObsCtx<MODE>& c, const ObsBTerm t, float pair_weight,
                                              v2f (&a)[2 * NP], v4f (&r)[NP], v2f& d2, uint32_t& cur_op, v2f (&dq)[4]) {
  constexpr bool ACC = MODE == OBS_LAMBDA || MODE == OBS_LAMBDA_VALUE;
  constexpr bool MULTI = MODE == OBS_VALUES_MULTI;
  const uint32_t op = t.meta & 1023u;
  float wv = t.coeff * pair_weight;
  if constexpr (MODE == OBS_LAMBDA) wv *= c.up[op];
  // sign at the thread's own index: block part and the (-1)^ny i^2 constant (scalar), thread part (one popcount); the
  // slot part is in the variant
  const uint32_t sg = uint32_t(__popc(c.bx & t.zb)) + ((t.meta >> 13) & 1u) + uint32_t(__popc(c.t & t.zt));
  const float wp = __uint_as_float(__float_as_uint(wv) ^ (sg << 31));
  const v2f w = v2f{wp, -wp};
  v2f s[4] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
  obs_term<ACC>(a, r, w, s, c.second ? t.off1 : t.off0, t.off1);
  if constexpr (!ACC) {
    const v2f inc = wp * ((s[0] + s[1]) + (s[2] + s[3]));
    if constexpr (MULTI) {
      // The first four observables keep a register accumulator each (four wave-uniform triangles: terms of
      // neighbouring masks alternate between observables -- XX, YY, XX, ... -- and a wave reduction per change costs
      // more than the term); the others share d2, reduced into this wave's LDS cell whenever the observable changes.
      if (op == 0u) dq[0] += inc;
      if (op == 1u) dq[1] += inc;
      if (op == 2u) dq[2] += inc;
      if (op == 3u) dq[3] += inc;
      if (op >= 4u) {
        if (op != cur_op) {
          if (cur_op != ~0u) {
            const float e = wave_sum(d2.x + d2.y);
            if ((c.tid & 63u) == 0u) c.cells[(c.tid >> 6) * c.n_ops + cur_op] += e;
          }
          d2 = v2f{0.f, 0.f};
          cur_op = op;
        }
        d2 += inc;
      }
    } else {
      d2 += inc;
    }
  }
}

template <int MODE, int BB>
__global__ __launch_bounds__(ObsShape<BB>::kThreads, 4) void observable_blocks_kernel(
    const float2* __restrict__ psi, float2* __restrict__ lam, uint32_t n, const ObsBTerm* __restrict__ terms,
    const ObsBGroup* __restrict__ groups, uint32_t n_groups, const float* __restrict__ upstream, uint32_t n_ops,
    uint32_t state0, float* __restrict__ value_part, uint32_t nb /* blocks per state */, uint32_t n_states,
    uint32_t xcd_states) {
  constexpr bool ACC = MODE == OBS_LAMBDA || MODE == OBS_LAMBDA_VALUE;
  constexpr bool HALVE = !ACC;
  constexpr bool MULTI = MODE == OBS_VALUES_MULTI;
  using Shape = ObsShape<BB>;
  constexpr bool HALVES = Shape::kHalves, BYTERMS = Shape::kByTerms, BYROWS = Shape::kByRows;
  constexpr int NP = int(Shape::kPairs), BITS = Shape::kBits;
  constexpr uint32_t kOT = Shape::kThreads, kOWaves = Shape::kWaves, kOBuf = Shape::kBuf, kOBlock = Shape::kBlock;
  extern __shared__ v4f lds4[];  // two block buffers of kOBuf 16-byte words; then the value cells
  float* cells = reinterpret_cast<float*>(lds4 + 2 * kOBuf);
  // (obs_rows reads absolute LDS addresses)
  if (uint32_t(uintptr_t((__attribute__((address_space(3))) char*)(lds4))) != 0u) __builtin_trap();
  // Workgroup -> (state, block).  Workgroups are dealt round-robin to the 8 XCDs (linear id mod 8), each with its own L2:
  //   xcd_states: XCD k works on state 8 g + k, its blocks in index order;
  //   otherwise XCD k takes the k-th contiguous eighth of every state.
  // Either way the workgroups an XCD runs at one time are neighbours in the index, walk the groups in the same
  // order, and fetch partner blocks from the same 2 MiB.
  uint32_t s_local, bx, pivot_mask = ~0u;
  {
    const uint32_t wg = blockIdx.x, per_group = 8u * nb, group = wg / per_group, r = wg - group * per_group;
    if (xcd_states && (group + 1u) * 8u <= n_states) {
      s_local = group * 8u + (r & 7u);
      bx = r >> 3;
    } else {
      if (!(nb & 7u) && nb > 8u) pivot_mask = (nb >> 3) - 1u;  // (obs_next_group: not the bits that select the XCD)
      const uint32_t q = r / nb, b = r - q * nb;
      s_local = group * 8u + q;
      bx = (nb & 7u) ? b : (b & 7u) * (nb >> 3) + (b >> 3);
    }
  }
  const uint32_t tid = threadIdx.x, t = tid & (kOH - 1u), hh = HALVES ? uni(tid >> 9) : 0u;
  const float2* ps = psi + (size_t(s_local) << n);
  const uint32_t own_row0 = BYROWS ? 4u * hh : 0u;   // the first row of the block this thread accumulates
  const v4f* own4 = reinterpret_cast<const v4f*>(ps + (size_t(bx) << BITS)) + t + 512u * own_row0;
  ObsCtx<MODE> c;
  c.terms = terms; c.lds = reinterpret_cast<const char*>(lds4); c.cells = cells; c.n_ops = n_ops; c.bx = bx; c.t = t; c.tid = tid;
  c.row_bytes = own_row0 << 13; c.second = BYROWS && hh;
  c.up = MODE == OBS_LAMBDA ? upstream + size_t(state0 + s_local) * n_ops : nullptr;

  v2f a[2 * NP];
  if constexpr (ACC) obs_zero_(a, std::make_integer_sequence<int, 2 * NP>{});
  else obs_own_(a, own4, std::make_integer_sequence<int, NP>{});
  if constexpr (MULTI) {
    for (uint32_t i = tid; i < kOWaves * n_ops; i += kOT) cells[i] = 0.f;
  }
  v4f r[NP];
  [[maybe_unused]] v4f r2[NP];   // (rows split between the halves: the partner rows of a SECOND mask, read with the first's)
  ObsSet pfa, pfb;
  obs_rows(r, c.lds, (t << 4) | c.row_bytes);  // (defined values before the first mask; never used)
  v2f d2 = v2f{0.f, 0.f};  // value modes: sum_k W_k (own . partner), both halves
  uint32_t cur_op = ~0u;  // (several observables: the one d2 is collecting, none yet)
  v2f dq[4] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
  const uint32_t row0 = 4u * hh;
  const uint32_t t16 = t << 4;
  auto stage_a = [&] { obs_stage(lds4 + t + 512u * row0, pfa); };
  auto stage_b = [&] { obs_stage(lds4 + kOBuf + t + 512u * row0, pfb); };
  // the block a group pairs this one with (past the last group: the block itself -- no branch around the registers)
  auto partner = [&](uint32_t g) { return ps + (size_t(bx ^ (g < n_groups ? groups[g].xout : 0u)) << BITS); };

  // groups of the next four steps
  uint32_t g0 = obs_next_group<HALVE>(groups, n_groups, 0u, bx, pivot_mask);
  uint32_t g1 = obs_next_group<HALVE>(groups, n_groups, g0 + 1u, bx, pivot_mask);
  uint32_t g2 = obs_next_group<HALVE>(groups, n_groups, g1 + 1u, bx, pivot_mask);
  // (plain vector loads before this point -- the block's own amplitudes in the value modes -- must have landed before
  // the manual counting starts)
  asm volatile("s_waitcnt vmcnt(0)");
  obs_fetch<BB>(pfa, partner(g0), t16, row0);
  obs_fetch<BB>(pfb, partner(g1), t16, row0);
  obs_wait_older<1>(pfa);
  stage_a();
  obs_fetch<BB>(pfa, partner(g2), t16, row0);
  __syncthreads();
  // Two steps per iteration (the prefetch sets alternate; a run-time choice between them would make the compiler copy
  // them).  Step: the block of g1 moves from its set to the buffer nobody reads in this step, the set is refilled
  // with the block of g3, the masks of g0 (this half's share) are applied from the other buffer; one barrier.
  auto terms_of = [&](const ObsBGroup gr, uint32_t cur_bytes) {
    const float pair_weight = (!ACC && gr.xout != 0u) ? 2.f : 1.f;
    uint32_t k = BYTERMS && hh ? gr.mid : gr.begin;
    const uint32_t k1 = BYTERMS && !hh ? gr.mid : gr.end;  // this half's masks (the other shapes: all of them)
    if constexpr (BYROWS && QHBM_OBS_ROWS_PAIRED) {
      // Four pairs per thread leave registers for a second set of partner rows: the rows of TWO masks are read together and
      // waited for once (a wave pays the LDS latency once per two terms).  A term on the mask of its predecessor re-reads
      // the rows when the predecessor's sat in the second set.
      if (k < k1) {
        ObsBTerm ta = terms[k];
        bool reread = false;
        for (;;) {
          const ObsBTerm tb = terms[k + 1u];
          const bool pair = k + 1u < k1 && (tb.meta & kObsNewMask);
          if ((ta.meta & kObsNewMask) || reread) obs_term_rows<MODE, NP>(c, ta, cur_bytes, r);
          if (pair) obs_term_rows<MODE, NP>(c, tb, cur_bytes, r2);
          obs_term_math<MODE, NP>(c, ta, pair_weight, a, r, d2, cur_op, dq);
          if (pair) {
            obs_term_math<MODE, NP>(c, tb, pair_weight, a, r2, d2, cur_op, dq);
            k += 2u;
            if (k >= k1) break;
            ta = terms[k];
            reread = true;
          } else {
            if (++k >= k1) break;
            ta = tb;
            reread = false;
          }
        }
      }
      return;
    }
    // two records in flight: the scalar load of the next one runs behind the arithmetic of the current one (the
    // array is padded by one record)
    if (k < k1) {
      ObsBTerm ta = terms[k];
      for (;;) {
        const ObsBTerm tb = terms[k + 1u];
        obs_one_term<MODE, NP>(c, ta, cur_bytes, pair_weight, a, r, d2, cur_op, dq);
        if (++k >= k1) break;
        ta = terms[k + 1u];
        obs_one_term<MODE, NP>(c, tb, cur_bytes, pair_weight, a, r, d2, cur_op, dq);
        if (++k >= k1) break;
      }
    }
  };
  // The halves take their two jobs of a step in OPPOSITE order: half 0 moves its rows of the next block into LDS and
  // then applies its masks, half 1 applies its masks first.  All sixteen waves storing at once is a burst the LDS
  // takes 830 cycles for (ds_write_b128: 79 B/clk) with the vector units idle; this way each half's stores run under
  // the other half's arithmetic.  (-DQHBM_OBS_SKEW=0: both halves store first, for A/B measurements.)
  const bool skew = QHBM_OBS_SKEW && HALVES && hh;
#ifdef QHBM_OBS_TIMING
  uint64_t tacc[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  auto refill_a = [&](uint32_t g3) {
    OBS_T(4)
    obs_wait_older<1>(pfa);
    OBS_T(0)
    stage_a();
    obs_fetch<BB>(pfa, partner(g3), t16, row0);
    OBS_T(1)
  };
  auto refill_b = [&](uint32_t g3) {
    OBS_T(4)
    obs_wait_older<1>(pfb);
    OBS_T(0)
    stage_b();
    obs_fetch<BB>(pfb, partner(g3), t16, row0);
    OBS_T(1)
  };
  while (g0 < n_groups) {
    {  // even step: masks of g0 from buffer 0; g1's block pfb -> buffer 1; pfb <- g3's block
      const ObsBGroup gr = groups[g0];
      const uint32_t g3 = obs_next_group<HALVE>(groups, n_groups, g2 + 1u, bx, pivot_mask);
      if (!skew) refill_b(g3);
      OBS_T(4)
      terms_of(gr, 0u);
      OBS_T(2)
      if (skew) refill_b(g3);
      OBS_T(4)
          __syncthreads();
      OBS_T(3)
      g0 = g1; g1 = g2; g2 = g3;
    }
    if (g0 >= n_groups) break;
    {  // odd step: masks of g0 from buffer 1; g1's block pfa -> buffer 0; pfa <- g3's block
      const ObsBGroup gr = groups[g0];
      const uint32_t g3 = obs_next_group<HALVE>(groups, n_groups, g2 + 1u, bx, pivot_mask);
      if (!skew) refill_a(g3);
      OBS_T(4)
      terms_of(gr, 8u * kOBlock);
      OBS_T(2)
      if (skew) refill_a(g3);
      OBS_T(4)
          __syncthreads();
      OBS_T(3)
      g0 = g1; g1 = g2; g2 = g3;
    }
  }
#ifdef QHBM_OBS_TIMING
  if ((blockIdx.x == 5000u || blockIdx.x == 20001u) && (tid & 63u) == 0u)
    printf("obs timing wg %u wave %2u: groups %u  wait %llu  stage+fetch %llu  terms %llu  barrier %llu  other %llu (x10 ns)\n", blockIdx.x, tid >> 6,
           n_groups, (unsigned long long)tacc[0], (unsigned long long)tacc[1], (unsigned long long)tacc[2], (unsigned long long)tacc[3], (unsigned long long)tacc[4]);
#endif
  obs_wait_older<0>(pfa);  // (the last refills re-read the block itself: nothing is pending past this point)
  obs_wait_older<0>(pfb);

  v4f* const buf0 = lds4;
  if constexpr (ACC) {
    if constexpr (BYTERMS) {
      // the second half's accumulators join the first half's through an LDS buffer (every wave is past its last read)
      if (hh) obs_acc_out_(buf0 + t, a, std::make_integer_sequence<int, 8>{});
      __syncthreads();
      if (!hh) obs_acc_in_(a, buf0 + t, std::make_integer_sequence<int, 8>{});
    }
    if ((!BYTERMS || !hh) && lam)
      obs_store_(reinterpret_cast<v4f*>(lam + (size_t(s_local) << n) + (size_t(bx) << BITS)) + t + 512u * own_row0, a,
                 std::make_integer_sequence<int, NP>{});
  }
  if constexpr (MODE == OBS_LAMBDA) return;
  if constexpr (MULTI) {
    if (cur_op != ~0u) {
      const float e = wave_sum(d2.x + d2.y);
      if ((tid & 63u) == 0u) cells[(tid >> 6) * n_ops + cur_op] += e;
    }
    {  // the register accumulators of observables 0..3 (cells nobody else of this wave touches any more)
      const float e0 = wave_sum(dq[0].x + dq[0].y), e1 = wave_sum(dq[1].x + dq[1].y);
      const float e2 = wave_sum(dq[2].x + dq[2].y), e3 = wave_sum(dq[3].x + dq[3].y);
      if ((tid & 63u) == 0u) {
        float* row = cells + (tid >> 6) * n_ops;
        row[0] += e0;
        if (n_ops > 1u) row[1] += e1;
        if (n_ops > 2u) row[2] += e2;
        if (n_ops > 3u) row[3] += e3;
      }
    }
    __syncthreads();
    for (uint32_t q = tid; q < n_ops; q += kOT) {
      float e = 0.f;
#pragma unroll
      for (uint32_t w8 = 0; w8 < kOWaves; ++w8) e += cells[w8 * n_ops + q];  // wave order: bit-reproducible
      value_part[(size_t(s_local) * nb + bx) * n_ops + q] = e;
    }
  } else {
    float e;
    if constexpr (MODE == OBS_LAMBDA_VALUE) {  // <psi|O|psi> = sum_j Re(conj(psi_j) lambda_j): the first half holds lambda
      e = (BYTERMS && hh) ? 0.f : obs_energy_(own4, a, std::make_integer_sequence<int, NP>{});
    } else {
      e = d2.x + d2.y;
    }
    e = wave_sum(e);
    __syncthreads();  // the buffers are free now
    if ((tid & 63u) == 0u) cells[tid >> 6] = e;
    __syncthreads();
    if (tid == 0u) {
      float tot = 0.f;
#pragma unroll
      for (uint32_t w8 = 0; w8 < kOWaves; ++w8) tot += cells[w8];
      value_part[size_t(s_local) * nb + bx] = tot;  // logical block: the sum order does not depend on the XCD map
    }
  }
}

// <psi|O_t|psi> of state s = sum of its blocks' partials, in block order (bit-reproducible), into the
// fixed-point value accumulator of (s, t).
__global__ __launch_bounds__(256) void value_parts_blocks_kernel(const float* __restrict__ value_part, uint32_t n_blocks,
                                                                 uint32_t n_ops, const float* __restrict__ op_scale,
                                                                 unsigned long long* __restrict__ out64, uint32_t state0) {
  __shared__ double part[256];
  const uint32_t s = blockIdx.x, t = blockIdx.y;
  double acc = 0.0;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 256u) acc += double(value_part[(size_t(s) * n_blocks + b) * n_ops + t]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (int(threadIdx.x) < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out64[size_t(state0 + s) * n_ops + t] += to_fixed(float(part[0]), op_scale[t]);
}

constexpr int kMaxDev = 64;
template <typename Kernel>
hipError_t obs_opt_in(Kernel kernel, bool (&done)[kMaxDev], size_t lds) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
  if (done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  if (e == hipSuccess) done[dev] = true;
  return e;
}

}  // namespace

size_t observable_blocks_value_parts(uint32_t n, uint32_t n_states, uint32_t n_ops, int block_bits) {
  return size_t(n_states) * (size_t(1) << (n - uint32_t(block_bits))) * std::max<uint32_t>(n_ops, 1u);
}

namespace {
template <int BB>
hipError_t launch_observable_blocks_shape(int mode, const float2* psi, float2* lam, uint32_t n, uint32_t n_states,
                                    const ObsBTerm* terms, const ObsBGroup* groups, uint32_t n_groups,
                                    const float* upstream, uint32_t n_ops, uint32_t state0, const float* op_scale,
                                    unsigned long long* out64, float* value_part, bool xcd_states, hipStream_t stream) {
  using Shape = ObsShape<BB>;
  constexpr uint32_t kOBlock = Shape::kBlock, kOWaves = Shape::kWaves, kOT = Shape::kThreads;
  if (n < uint32_t(Shape::kBits) || n_states == 0) return n_states ? hipErrorInvalidValue : hipSuccess;
  if (mode == OBS_VALUES_MULTI && n_ops > kObsMaxValueOps) return hipErrorInvalidValue;
  const uint32_t nb = 1u << (n - uint32_t(Shape::kBits));
  // (a state must at least fill an XCD's workgroup slots: 32 CUs x the workgroups of a CU)
  const uint32_t xs = xcd_states && nb >= (Shape::kHalves ? 64u : 128u) ? 1u : 0u;
  // the largest cell area any mode uses, so that every instantiation is opted in once for the same size
  const size_t lds = 2u * size_t(kOBlock) * 8u + size_t(kOWaves) * kObsMaxValueOps * sizeof(float);
  static bool done[4][kMaxDev];
  hipError_t e = hipSuccess;
#define QHBM_OBSB(M_)                                                                                                   \
  {                                                                                                                    \
    e = obs_opt_in(observable_blocks_kernel<M_, BB>, done[M_], lds);                                                    \
    if (e != hipSuccess) return e;                                                                                     \
    const size_t use = 2u * size_t(kOBlock) * 8u + (M_ == OBS_VALUES_MULTI ? size_t(kOWaves) * n_ops * sizeof(float) : 64u); \
    hipLaunchKernelGGL((observable_blocks_kernel<M_, BB>), dim3(nb * n_states), dim3(kOT), use, stream, psi, lam, n, terms, \
                       groups, n_groups, upstream, n_ops, state0, value_part, nb, n_states, xs);                       \
  }
  switch (mode) {
    case OBS_LAMBDA: QHBM_OBSB(OBS_LAMBDA) break;
    case OBS_LAMBDA_VALUE: QHBM_OBSB(OBS_LAMBDA_VALUE) break;
    case OBS_VALUES: QHBM_OBSB(OBS_VALUES) break;
    case OBS_VALUES_MULTI: QHBM_OBSB(OBS_VALUES_MULTI) break;
    default: return hipErrorInvalidValue;
  }
#undef QHBM_OBSB
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (mode != OBS_LAMBDA) {
    const uint32_t T = mode == OBS_VALUES_MULTI ? n_ops : 1u;
    hipLaunchKernelGGL(value_parts_blocks_kernel, dim3(n_states, T), dim3(256), 0, stream, value_part, nb, T, op_scale,
                       out64, state0);
    e = hipGetLastError();
  }
  return e;
}
}  // namespace

hipError_t launch_observable_blocks(int mode, int block_bits, const float2* psi, float2* lam, uint32_t n, uint32_t n_states,
                                    const ObsBTerm* terms, const ObsBGroup* groups, uint32_t n_groups,
                                    const float* upstream, uint32_t n_ops, uint32_t state0, const float* op_scale,
                                    unsigned long long* out64, float* value_part, bool xcd_states, hipStream_t stream) {
  if (block_bits == kObsBlockBits)
    return launch_observable_blocks_shape<kObsBlockBits>(mode, psi, lam, n, n_states, terms, groups, n_groups, upstream, n_ops,
                                                         state0, op_scale, out64, value_part, xcd_states, stream);
  if (block_bits == kObsBlockBitsSmall)
    return launch_observable_blocks_shape<kObsBlockBitsSmall>(mode, psi, lam, n, n_states, terms, groups, n_groups, upstream,
                                                              n_ops, state0, op_scale, out64, value_part, xcd_states, stream);
  if (block_bits == kObsShapeRows)
    return launch_observable_blocks_shape<kObsShapeRows>(mode, psi, lam, n, n_states, terms, groups, n_groups, upstream, n_ops,
                                                         state0, op_scale, out64, value_part, xcd_states, stream);
  return hipErrorInvalidValue;
}

}  // namespace qhbm
