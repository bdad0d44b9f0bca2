// kernels.h -- launch entry points of kernels.hip (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "program.h"
#include "schedule.h"

namespace qhbm {

struct DevTerm {  // one Pauli term, amplitude-index bit space
  float coeff;
  uint32_t x, z;
  uint32_t ny;
  uint32_t op;
};

// Terms of equal x mask, consecutive in the x-sorted term array: [previous end, end).  A group
// never straddles a multiple of kObsTermChunk / 2 (the kernel stages that many terms in LDS at a time).
// Within a group of real weights the terms are ordered by how their sign (-1)^{popc(j & z)} varies
// inside a workgroup of apply_observable_kernel (j = block | slot bits | tid << 1: a thread owns A / 2
// adjacent pairs, slot s = index bit 0 and the bits from 9 up): first the n_h terms whose z misses the
// thread bits (one pre-summed weight per slot), then the n_l terms whose z misses the slot bits (one
// signed sum per thread), then the rest (per amplitude).
struct ObsGroup {
  uint32_t x, end;
  uint32_t has_imag;  // some term of the group has an odd number of Y factors (imaginary weight)
  uint32_t n_h, n_l;
  // Several observables with a value accumulator each (apply_observable_kernel<A, OBS_GATHER_MULTI>, at most
  // kObsGatherMultiOps of them): the terms are sorted by (x, observable), a group holds ONE observable's terms of one mask,
  // and a group whose mask equals the previous group's re-uses its gathered partners (same_x bit 0; bit 1: the array is
  // in that order at all -- no group may then skip partner pairs as vanishing, its successor may need them).
  uint32_t op, same_x;
};
enum ObsGatherMode : int {
  OBS_GATHER_LAMBDA = 0,  // lambda = sum_k upstream[s, op_k] c_k P_k psi
  OBS_GATHER_VALUE = 1,   // one observable: lambda = O psi (unweighted) and <psi|O|psi>
  OBS_GATHER_MULTI = 2,   // 2..kObsGatherMultiOps observables: the weighted lambda (if asked for) AND every <psi|O_t|psi>
};
constexpr uint32_t kObsGatherMultiOps = 4;
// A far launch of the two-level lambda = O psi sweep (engine.cpp far_windows, kernels.hip obs_phys): the terms whose masks
// flip only index bits [far_hi, far_hi + 7) (and bits 0..3), with x and z given in the VIRTUAL index space in which that
// window has changed places with bits [4, 11).
struct ObsFarLaunch {
  const DevTerm* terms;
  uint32_t n_terms;
  const ObsGroup* groups;
  uint32_t n_groups;
  uint32_t far_hi;
};
constexpr uint32_t obs_amps_per_thread(uint32_t n) { return n >= 11 ? 8u : 4u; }  // A of apply_observable_kernel<A>
constexpr uint32_t kObsThreadMask = 0x1feu;                                        // index bits 1..8 = the thread
constexpr uint32_t obs_slot_mask(uint32_t n) { return 1u | ((obs_amps_per_thread(n) / 2u - 1u) << 9); }  // bit 0 and 9 (, 10)
constexpr uint32_t kObsTermChunk = 512;  // 2 x the terms staged in LDS at a time (13 KiB: five workgroups per CU)

// ---- block-grouped Pauli sums (observable.hip) ---------------------------------------------------
// lambda = O psi and <psi|O|psi> for operators with MANY X-masks (BASELINE config 4: 512 random strings, 480 masks).
// A workgroup owns a BLOCK of 2^kObsBlockBits consecutive amplitudes.  A mask x = x_out (block bits) | x_in (bits inside
// the block) pairs the block with the partner block b ^ x_out: the terms come sorted by x_out and cut into GROUPS of
// equal x_out, the partner block of a group is fetched ONCE into LDS and every mask of the group reads it from there at
// l ^ x_in (config 4: 173 fetches per block instead of 453 gathers).
// Two shapes of the same kernel (observable.hip): blocks of 2^13 under ONE workgroup of 1024 threads per CU whose two halves
// split a group's masks, or blocks of 2^12 under TWO independent workgroups of 512 threads per CU that apply every mask of
// their groups (engine option "observable_block_bits").
constexpr int kObsBlockBits = 13;      // the larger shape: what the engine requires of a state before it picks these kernels
constexpr int kObsBlockBitsSmall = 12;
constexpr int kObsShapeRows = 113;     // launch_observable_blocks `block_bits`: blocks of 2^13 (the tables of 13), the halves split the ROWS
constexpr uint32_t kObsNewMask = 1u << 12;   // ObsBTerm::meta: the term's x differs from the previous term's
constexpr uint32_t kObsMaxValueOps = 256;    // per-op value cells of a workgroup (one row per wave: 16 KiB) must fit LDS
struct ObsBTerm {  // 32 bytes, one s_load_dwordx8: everything the kernel would otherwise derive per term with scalar ALU work
  float coeff;
  uint32_t zt;     // Z mask on the thread bits: (z >> 1) & 511
  uint32_t zb;     // Z mask on the block bits: z >> block bits
  uint32_t xrow;   // byte-address XOR of the partner rows in the LDS buffer: ((x >> 1) & 511) << 4 | ((x >> 10) & 7) << 13
                   // (blocks of 2^12: four rows, (x >> 10) & 3)
  uint32_t off0;   // jump-table offset of the variant for slots 0..7 (observable_variants.inc) ...
  uint32_t off1;   // ... and for slots 8..15 (base sign flipped when z holds the highest slot bit; blocks of 2^12: unused)
  uint32_t meta;   // op (bits 0..9) | kObsNewMask | kObsSignBit when i^ny (-1)^ny contributes a minus sign
  uint32_t pad;
};
constexpr uint32_t kObsSignBit = 1u << 13;
constexpr uint32_t kObsChunkBytes = 68, kObsPreambleBytes = 12;  // layout of the variant tables (scripts/gen_observable_asm.py)
struct ObsBGroup {
  uint32_t xout;  // x >> block bits: partner block = block ^ xout
  uint32_t begin, end;  // terms [begin, end) ...
  uint32_t mid;         // ... of which [begin, mid) are the first half-workgroup's masks, [mid, end) the second's (blocks of 2^12: = end)
};
enum ObsBlocksMode : int {
  OBS_LAMBDA = 0,        // lambda = sum_k upstream[s, op_k] c_k P_k psi
  OBS_LAMBDA_VALUE = 1,  // one observable: lambda = O psi (unweighted) and <psi|O|psi>
  OBS_VALUES = 2,        // one observable: <psi|O|psi> only (nothing stored)
  OBS_VALUES_MULTI = 3,  // n_ops <= kObsMaxValueOps observables: <psi|O_t|psi> for every t (nothing stored)
};
// value modes: value_part holds observable_blocks_value_parts(n, n_states, n_ops) floats of scratch
size_t observable_blocks_value_parts(uint32_t n, uint32_t n_states, uint32_t n_ops, int block_bits);
// `block_bits`: kObsBlockBits or kObsBlockBitsSmall -- the one the tables `terms` / `groups` were built for
hipError_t launch_observable_blocks(int mode, int block_bits, const float2* psi, float2* lam, uint32_t n, uint32_t n_states,
                                    const ObsBTerm* terms, const ObsBGroup* groups, uint32_t n_groups,
                                    const float* upstream, uint32_t n_ops, uint32_t state0, const float* op_scale,
                                    unsigned long long* out64, float* value_part, bool xcd_states, hipStream_t stream);

size_t fwd_lds_bytes(int K);
size_t adj_lds_bytes(int K, bool exchange);

// op_scale[k] = 2^(kValueFracBits - ceil(log2 sum|c|)) of op k; out64 [U, n_ops] fixed-point accumulators.
hipError_t launch_pass_fwd(int K, int R, const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits,
                           int n_user, const uint32_t* prog, const uint32_t* tables, const float* coef,
                           const float* op_scale, unsigned long long* out64, uint32_t state0, hipStream_t stream,
                           const float2* psi_src = nullptr /* batched programs: element e LOADS the tile of state
                           e % prog_states of this buffer and stores into its own element of psi */);
// The same pass on PAIRS of consecutive states, tile pair in registers (kernels.hip pass_fwd2_kernel): only for
// lean passes that prune nothing, load nothing stale and measure nothing (or whose measurements are ignored).
bool pass_fwd_pair_supported(int K);
hipError_t launch_pass_fwd_pair(int K, const PassArgs& a, uint32_t n_states, float2* psi, const int8_t* bits, int n_user,
                                uint32_t state0, const uint32_t* prog, const uint32_t* tables, const float* coef,
                                hipStream_t stream);
// p[0 .. bytes) = 0 by a kernel (not a memset node: see kernels.hip)
hipError_t launch_zero_fill(void* p, size_t bytes, hipStream_t stream);
hipError_t launch_values_from_fixed(const unsigned long long* acc, const float* inv_scale, float* out, uint32_t count,
                                    uint32_t n_ops, hipStream_t stream);
// tile_grad [n_states * tiles, a.n_slots]: one gradient row per workgroup.  `exchange` selects the
// register-resident tile pair with one LDS exchange buffer (lean programs only).
hipError_t launch_pass_adj(int K, bool exchange, const PassArgs& a, uint32_t n_states, float2* psi, float2* lam,
                           const int8_t* bits, int n_user, const uint32_t* prog, const uint32_t* tables,
                           const float* coef, float* tile_grad, uint32_t state0, hipStream_t stream);
// tile_grad must have room for reduce_tiles_scratch_rows(n_states, n_tiles) more rows of n_slots floats.
size_t reduce_tiles_scratch_rows(size_t n_states, size_t n_tiles);
hipError_t launch_reduce_tiles(float* tile_grad, uint32_t n_states, uint32_t n_tiles, uint32_t n_slots,
                               float* state_grad, uint32_t n_slots_total, uint32_t slot_base, uint32_t state0,
                               hipStream_t stream);
// (value mode, out64 != null: `value_part` holds observable_value_parts(n, n_states) floats of scratch)
size_t observable_value_parts(uint32_t n, uint32_t n_states);
// (out64 != null: the values go to the fixed-point accumulators -- one observable, or `multi` (terms and groups in the
// (x, observable) order of OBS_GATHER_MULTI; value_part then holds observable_value_parts(n, n_states) x n_ops floats))
hipError_t launch_apply_observable(const float2* psi, float2* lam, uint32_t n, uint32_t n_states,
                                   const DevTerm* terms, uint32_t n_terms, const ObsGroup* groups,
                                   uint32_t n_groups, const float* upstream, uint32_t n_ops, uint32_t state0,
                                   const float* op_scale, unsigned long long* out64, float* value_part,
                                   bool xcd_states, hipStream_t stream, bool multi = false,
                                   const struct ObsFarLaunch* far = nullptr, int n_far = 0);
// Terms measured on the final state in HBM (X-mask wider than a tile); accumulates into out64.
hipError_t launch_measure_global(const float2* psi, uint32_t n, uint32_t n_states, const DevTerm* terms,
                                 uint32_t n_terms, const float* op_scale, unsigned long long* out64, uint32_t n_ops,
                                 uint32_t state0, hipStream_t stream);
hipError_t launch_parity_energy(const int8_t* bits, int64_t n_rows, int n, const uint64_t* masks,
                                const float* thetas, int n_terms, float* energy, hipStream_t stream);
hipError_t launch_parity_energy_vjp(const int8_t* bits, int64_t n_rows, int n, const uint64_t* masks, int n_terms,
                                    const float* w, float* grad, hipStream_t stream);
// block_cum: n_states * 2^n / 1024 doubles of scratch.
hipError_t launch_sample(const float2* psi, uint32_t n, int n_user, uint32_t n_states, double* block_cum,
                         uint32_t n_shots, uint64_t seed, uint32_t state0, int8_t* out, hipStream_t stream);
// One gate with a non-zero cirq global_shift: phase exp(i pi shift (scalar * params[param_idx] + offset)).
struct ShiftPhase {
  int32_t param_idx;
  float scalar, offset, shift;
};
// Shot counts per outcome for n_elements = programs x prog_states (program, state) elements whose final
// states sit consecutively in psi (element e = program prog0 + e / prog_states on state state0 + e % prog_states):
// out [programs, n_states_total, 2^n_user] int32 (zeroed by the caller for n > 10).  block_cum: n_elements * 2^n / 1024 doubles.
hipError_t launch_sample_counts(const float2* psi, uint32_t n, int n_user, uint32_t n_elements, uint32_t prog_states,
                                uint32_t prog0, double* block_cum, uint32_t n_shots, uint64_t seed, uint32_t state0,
                                uint32_t n_states_total, int* out, hipStream_t stream);
hipError_t launch_global_phase(const CoefJob* jobs, int n_jobs, const ShiftPhase* shifts, int n_shifts,
                               const float* params, float* out_cs, hipStream_t stream);
hipError_t launch_scale_states(float2* st, size_t count, const float* cs, hipStream_t stream);
hipError_t launch_prep_coefs(const CoefJob* jobs, int n_jobs, const float* params, float* coef,
                             int shift_gate, double shift, hipStream_t stream);
// n_programs coefficient buffers coef_stride floats apart (1, 0 for the usual single program)
hipError_t launch_combine_diag(float* coef, const uint32_t* rec_offsets, int n_records, uint32_t n_programs,
                               uint32_t coef_stride, hipStream_t stream);
// Batched parameter-shift programs: program y shifts the exponent of gate shift_gates[y] by shifts[y].
hipError_t launch_prep_coefs_batch(const CoefJob* jobs, int n_jobs, const float* params, float* coef,
                                   const int* shift_gates, const float* shifts, uint32_t n_programs,
                                   uint32_t coef_stride, hipStream_t stream);
hipError_t launch_replicate(const float* src, float* dst, uint32_t words, uint32_t stride, uint32_t copies,
                            hipStream_t stream);
hipError_t launch_shift_program_accumulate(const float* vals, const float* upstream, uint32_t n_programs, uint32_t c,
                                           uint32_t n_ops, uint32_t s0, double* prog_acc, hipStream_t stream,
                                           const int* dst_index = nullptr /* program q adds to prog_acc[dst_index[q]] */);
hipError_t launch_shift_combine(const double* prog_acc, const int* gate_param, const float* gate_weight,
                                int n_shift_gates, float* grad, int n_params, hipStream_t stream);
hipError_t launch_reduce_grad(const float* state_grad, uint32_t U, uint32_t n_slots,
                              const int* param_slot_begin, const int* param_slots,
                              const float* slot_factor, float* grad, int n_params, int accumulate,
                              hipStream_t stream);
hipError_t launch_scale_rows(float* rows, uint32_t U, uint32_t width, const float* w, hipStream_t stream);
// sustained packed-fp32 issue rate and shader clock (engine.cpp qhbm_clock_probe)
hipError_t launch_clock_probe(uint64_t* out, float* sink, uint32_t n_cus, hipStream_t stream);
uint32_t clock_probe_waves(uint32_t n_cus);
double clock_probe_instructions_per_simd();
hipError_t launch_scatter_jac(const float* state_grad, uint32_t U, uint32_t n_slots,
                              const int* param_slot_begin, const int* param_slots,
                              const float* slot_factor, float* jac, uint32_t n_ops, uint32_t op,
                              uint32_t n_params, hipStream_t stream);


}  // namespace qhbm
