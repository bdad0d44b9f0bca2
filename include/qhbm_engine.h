/*
 * qhbm_engine.h -- C ABI of the MI355X-native expectation engine.
 *
 * This is the drop-in boundary for the hot path of google/qhbm-library:
 *
 *   qhbmlib/inference/qnn.py:82-84   QuantumInference._expectation(circuits,
 *       symbol_names, symbol_values, observables)            <- qhbm_expectation
 *   qhbmlib/inference/qnn.py:134-138 tfq.layers.Expectation()(circuits,
 *       symbol_names, symbol_values, operators) forward       <- qhbm_expectation
 *   qhbmlib/inference/qnn.py:90-99   adjoint backward of that op (VJP w.r.t.
 *       symbol_values, summed through the tile of qnn.py:75-76) <- qhbm_expectation_vjp
 *   qhbmlib/inference/qnn.py:168,192-194 ParameterShift gradient circuits
 *       (rule restated in baselines/train.py:190-240)          <- qhbm_expectation_vjp(method=1)
 *   qhbmlib/models/circuit.py:129-136 bit-injection X**bit     <- `bits` argument
 *   qhbmlib/models/circuit.py:138-178 append / inverse         <- gate list passed to qhbm_set_circuit
 *
 * Conventions
 *   - qubit q in [0, n_qubits). Qubit 0 is the MOST significant bit of a basis
 *     index (cirq big-endian; qhbmlib/inference/ebm.py:445-447).
 *   - `bits` is row-major [U, n_qubits] int8; column j initialises qubit j.
 *   - exponent of a gate: t = scalar * params[param_idx] + offset
 *     (param_idx < 0: t = offset).  Inverse circuit = reversed gate list with
 *     scalar and offset negated (circuit.py:164-176).
 *   - Pauli masks are in QUBIT space: bit q of x_mask set  <=> X or Y on qubit q,
 *     bit q of z_mask set <=> Z or Y on qubit q.
 *   - All pointers named d_* are DEVICE pointers (HBM); the others are host
 *     pointers.  Work is enqueued on `stream` (a hipStream_t cast to void*;
 *     NULL = default stream) and is asynchronous with respect to the host.
 *   - Every function returns 0 on success, non-zero on error;
 *     qhbm_last_error() gives the message.  No exceptions cross the ABI.
 *   - One engine per device; an engine is not thread-safe, distinct engines are
 *     independent.
 */
#ifndef QHBM_ENGINE_H_
#define QHBM_ENGINE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QHBM_ABI_VERSION 5

/* Gate kinds: the one-parameter "power gate" families of cirq 0.14.1 that
 * tensorflow-quantum 0.6.1 serialises.  A gate is
 * G**t with the cirq matrix convention
 *     G**t = sum_k exp(i*pi*t*e_k) P_k      (eigen-exponents e_k, projectors P_k)
 * times cirq's global phase exp(i*pi*t*global_shift) (qhbm_gate::global_shift; 0 for the
 * plain power gates).  rx/ry/rz(theta) are XPOW/YPOW/ZPOW with scalar = 1/pi and
 * global_shift = -0.5.  The phase never changes an expectation value, a gradient or a
 * sample, so the hot path ignores it; qhbm_statevector applies the product of all gates'
 * phases to the exported states, which therefore equal cirq's final_state_vector -- and the
 * matrix tfq.layers.Unitary returns (qhbmlib/inference/qnn_utils.py:23-33) -- exactly, not up
 * to a phase.  PhasedXPow, FSim and PhasedISwapPow are lowered by the host to products of
 * these kinds. */
enum qhbm_gate_kind {
  QHBM_GATE_I = 0,
  QHBM_GATE_XPOW = 1,
  QHBM_GATE_YPOW = 2,
  QHBM_GATE_ZPOW = 3,
  QHBM_GATE_HPOW = 4,
  QHBM_GATE_CZPOW = 5,
  QHBM_GATE_CNOTPOW = 6,  /* q0 = control, q1 = target */
  QHBM_GATE_SWAPPOW = 7,
  QHBM_GATE_ISWAPPOW = 8,
  QHBM_GATE_XXPOW = 9,
  QHBM_GATE_YYPOW = 10,
  QHBM_GATE_ZZPOW = 11,
  QHBM_GATE_KIND_COUNT = 12
};

typedef struct qhbm_gate {
  int32_t kind;      /* enum qhbm_gate_kind */
  int32_t q0;        /* first qubit */
  int32_t q1;        /* second qubit, -1 for one-qubit gates */
  int32_t param_idx; /* index into params[], -1 = constant exponent */
  float scalar;      /* exponent = scalar * params[param_idx] + offset */
  float offset;
  float global_shift; /* cirq EigenGate global_shift: the gate is exp(i*pi*t*global_shift) * G**t
                         (ABI v3; cirq.rx/ry/rz and tfq.util.exponential's
                         rotations -- qhbmlib/models/circuit.py:268-272 -- carry -0.5) */
} qhbm_gate;

/* Gradient method for qhbm_expectation_vjp. */
enum qhbm_grad_method {
  QHBM_GRAD_ADJOINT = 0,        /* adjoint sweep, what tfq.layers.Expectation uses */
  QHBM_GRAD_PARAMETER_SHIFT = 1 /* two shifted forwards per gate occurrence */
};

typedef struct qhbm_engine qhbm_engine;

/* ---- lifetime ---------------------------------------------------------- */
int qhbm_abi_version(void);
/* Creates an engine bound to HIP device `device`. */
int qhbm_create(int device, qhbm_engine** out);
void qhbm_destroy(qhbm_engine* h);
/* Message of the last failing call on this engine (or of qhbm_create when
 * h == NULL).  Valid until the next call. */
const char* qhbm_last_error(const qhbm_engine* h);

/* ---- model ------------------------------------------------------------- */
/* Installs the total circuit (bit-injector excluded: it is the `bits`
 * argument).  Replaces tfq serialisation + append_circuit
 * (circuit.py:129-160).  The gate list is copied. */
int qhbm_set_circuit(qhbm_engine* h, int n_qubits, int n_gates,
                     const qhbm_gate* gates, int n_params);

/* Which parameters want a gradient (optional; after qhbm_set_circuit, which resets it).  needs_grad[p] == 0 freezes
 * parameter p: its entries of every gradient the engine returns are 0, its gates get no gradient work (no slot in the
 * adjoint sweep, no shifted programs under the parameter-shift rule), and the adjoint sweep STOPS at the first gate,
 * in circuit order, of a parameter that is not frozen -- the leading gates of frozen parameters are never un-applied.
 * The reference has no such notion (tfq's differentiators return d/d(symbol) for every symbol, qnn.py:75-76, used or
 * not); the host mirror derives the mask from torch's requires_grad of the circuits that make up the total circuit
 * (QMHL with a fixed data QHBM: the data circuit's half).  NULL restores "all".  Host pointer, n_params entries. */
int qhbm_set_gradient_mask(qhbm_engine* h, const uint8_t* needs_grad, int n_params);

/* Installs n_ops observables; op k is sum_{j in [term_offsets[k],
 * term_offsets[k+1])} coeffs[j] * Pauli(x_masks[j], z_masks[j]).
 * Replaces the tiled PauliSum protos of qnn.py:133.  Arrays are copied. */
int qhbm_set_observables(qhbm_engine* h, int n_ops, const int32_t* term_offsets,
                         const float* coeffs, const uint64_t* x_masks,
                         const uint64_t* z_masks);

/* ---- tuning ------------------------------------------------------------ */
/* Optional knobs (name -> value); unknown names are an error.
 *   "tile_qubits"          log2 amplitudes of one LDS tile (10..14), 0 = auto
 *   "adjoint_tile_qubits"  the same for the backward sweep (10..13), 0 = auto (12, or 13 when that plan's
 *                          modelled time -- arithmetic of qhbm_flop_model, tile traffic -- is at least 2 % lower)
 *   "chunk_states"         states simulated per launch group, 0 = auto
 *   "workspace_budget_mb"  cap on the statevector workspace; 0 = a third of the device's memory
 *   "profile_events"       record HIP events around the pass kernels (qhbm_kernel_time_ms)
 * Developer knobs for A/B measurements (defaults are the measured best): "measure_tile_qubits"
 * (tile of measurement-only passes, 0 = largest), "adjoint_exchange" (1 = register-resident tile
 * pair with one LDS exchange buffer, 0 = both tiles in LDS), "full_diag_threshold" /
 * "adjoint_full_diag_threshold", "round_qubits" (must be 4), "force_general_kernels",
 * "cph_wave_bits" (1 = the scheduler's layout choices of round 2: the partner bits of boundary
 * controlled phases and the bits with no gate left become wave bits, so that whole waves skip work
 * on predicates that are off / on zeros of psi, and adjoint tail passes use tiles without the low
 * index bits; 0 = the plain layout, for A/B measurements), "values_from_observable" (1 = with a single
 * observable the expectation value is taken from lambda = O psi in the calls that compute lambda
 * anyway, and the forward sweep measures nothing; 0 = always measure in the forward sweep),
 * "adjoint_stop_early" (with frozen parameters, qhbm_set_gradient_mask: 1 = the backward sweep stops at the first
 * gate of a live parameter, 0 = it runs to the basis state and prunes its tail, -1 = whichever the time model prefers),
 * "adjoint_plan_search" (1 = the backward plan is the one with the least modelled time among the pass orders the
 * scheduler ranked best and the greedy order; 0 = the scheduler's first choice),
 * "forward_values_from_observable" (forward-only calls with a single observable: -1 = the same kernel, storing
 * nothing, supplies the value when the plan has more than one pass and some term flips two or more qubits or
 * needs a measurement-only pass; 0 = measure in the passes; 1 = always),
 * "adjoint_relabel" (1 = adjoint plans move finished index bits out of the 128-byte lines when the finishing
 * pass stores its tiles), "forward_pairs" (1 = dense lean forward passes run on pairs of states with the
 * tiles in registers), "wide_last_pass" (-1 = the last forward gate pass may take a tile one or two bits
 * wider when that saves a pass, unless tile_qubits is set; 0 = never; 1 = always), "observable_xcd_states"
 * (lambda = O psi: 1 = one state per XCD at a time, 0 = every XCD an eighth of each state, -1 = by state size: states of
 * 64 MiB and more are shared), "observable_kernel" (lambda = O psi and <psi|O|psi>: 0 = one L2 gather per X-mask and
 * block of 2^11 amplitudes, 1 = partner blocks of 2^13 amplitudes staged in LDS once per group of masks that share them
 * (states of >= 13 qubits), -1 = whichever a fitted cost model prefers: the block kernel for operators with many masks),
 * "observable_block_bits" (shape of that block kernel: 13 (default) = blocks of 2^13 amplitudes under ONE workgroup of 1024
 * threads per CU whose two halves split a group's masks; 12 = blocks of 2^12 under TWO independent workgroups of 512 threads
 * per CU, every mask of a group applied by the one workgroup -- measured SLOWER, 94.4 against 51.2 ms for lambda on BASELINE
 * config 4: the co-resident workgroups drift apart and the window of partner blocks leaves the L2, hit rate 0.27 against
 * 0.77; kept for A/B runs), "observable_split_rows" (blocks of 2^13: 1 = the two halves of the workgroup split the block's
 * ROWS and each applies every mask of a group, instead of splitting the group's masks -- measured SLOWER, 57.6 against 51.3 ms:
 * twice the instructions per term; 0 = default),
 * "multi_observable_values" (several observables: -1 = their values come from ONE launch of the block kernel over the
 * final states, after lean measurement-free passes, when some term flips two or more qubits and there are at most 64
 * observables; 0 = always measured in the passes; 1 = always from the kernel, up to 256 observables),
 * "gather_multi_values" (2..4 observables: 1 = the gather kernel forms lambda AND carries a value accumulator per
 * observable -- one launch instead of two; 0 (default) = the values from the block kernel's own launch: the one-launch
 * form measured SLOWER, 103 against 64.7 ms on BASELINE config 3 split into its XX / YY / ZZ sums),
 * "observable_far_windows" (gather kernel, one observable or lambda alone: the masks that flip only bits of a seven-bit
 * window of far index bits (and bits 0..3) are applied by an extra launch per window that works in the index space with
 * that window swapped into bits 4..10 -- no partner run of theirs crosses the fabric.  0 (default) = never: measured NOT
 * faster at 28 qubits (65.0 ms in one launch, 66.2 in three); 1 = always, any window that holds a mask; -1 = from 26
 * qubits up, windows with three masks or more).
 */
int qhbm_set_option(qhbm_engine* h, const char* name, int64_t value);

/* Bytes of device workspace the engine will hold for a batch of U states
 * (forward only, or forward + adjoint when with_vjp != 0). */
int qhbm_workspace_bytes(qhbm_engine* h, int U, int with_vjp, size_t* out);

/* Bytes of device memory the engine holds right now (statevector workspace and gradient
 * partials); a host-side cache of engines uses it to bound its total footprint. */
int qhbm_allocated_bytes(qhbm_engine* h, size_t* out);

/* ---- hot path ---------------------------------------------------------- */
/* out[u, k] = <x_u| C(params)^dagger  O_k  C(params) |x_u>
 *   d_bits   [U, n_qubits] int8   (device)
 *   d_params [n_params]    float  (device)
 *   d_out    [U, n_ops]    float  (device) */
int qhbm_expectation(qhbm_engine* h, const int8_t* d_bits, int U,
                     const float* d_params, float* d_out, void* stream);

/* Values plus one vector-Jacobian product:
 *   d_grad[p] = sum_{u,k} d_upstream[u,k] * d out[u,k] / d params[p]
 *   d_upstream [U, n_ops] float, d_out_vals [U, n_ops] float (may be NULL),
 *   d_grad [n_params] float (overwritten). */
int qhbm_expectation_vjp(qhbm_engine* h, const int8_t* d_bits, int U,
                         const float* d_params, const float* d_upstream,
                         float* d_out_vals, float* d_grad, int method,
                         void* stream);

/* The same pair as two calls for an autograd-style caller (forward now, backward later, as
 * tf.custom_gradient / torch.autograd.Function drive qnn.py:134-138): qhbm_expectation_retain
 * is qhbm_expectation but leaves the final states in the workspace when the batch fits one
 * backward chunk; qhbm_expectation_vjp_retained then runs only lambda = O psi and the backward
 * sweep on them (same bits and params as the retaining call) and consumes them.  It fails if
 * nothing is retained -- any other compute call, or a batch too large, drops the states -- and
 * the caller falls back to qhbm_expectation_vjp. */
int qhbm_expectation_retain(qhbm_engine* h, const int8_t* d_bits, int U,
                            const float* d_params, float* d_out, void* stream);
int qhbm_expectation_vjp_retained(qhbm_engine* h, const int8_t* d_bits, int U,
                                  const float* d_params, const float* d_upstream,
                                  float* d_grad, void* stream);
/* Number of states the workspace currently retains for qhbm_expectation_vjp_retained: U right
 * after a qhbm_expectation_retain that kept its states, 0 otherwise (batch larger than one
 * backward chunk, or any compute call since). */
int qhbm_retained_states(qhbm_engine* h, int* out_U);

/* Per-state rows of the last adjoint VJP (qhbm_expectation_vjp with method 0, or
 * qhbm_expectation_vjp_retained) on U states:
 *   d_rows[u, p] = sum_k d_upstream[u, k] * d out[u, k] / d params[p]       [U, n_params] float
 * so that sum_u d_rows[u, :] is that call's d_grad.  A multi-GPU host gathers the rows of every
 * rank and adds them in global state order: the [P] gradient is then bit-identical for any number
 * of ranks (SURVEY.md 8e "fixed-order summation"); an all-reduce of d_grad is the fast path. */
int qhbm_state_gradients(qhbm_engine* h, int U, float* d_rows, void* stream);

/* Full Jacobian d_jac[u, k, p] (tests / small n only; adjoint). */
int qhbm_expectation_jacobian(qhbm_engine* h, const int8_t* d_bits, int U,
                              const float* d_params, float* d_out_vals,
                              float* d_jac, void* stream);

/* Final state vectors C(params)|x_u> (the data behind
 * qhbmlib/inference/qnn_utils.py:23-33 `unitary` (tfq.layers.Unitary) and
 * qhbm_utils.py:24-116 `density_matrix` / `fidelity`).
 *   d_out_states [U, 2^n_qubits] complex64 as interleaved (re, im) floats (device);
 *   amplitude index = the bitstring read as a big-endian binary number (qubit 0 is
 *   the most significant bit, as cirq orders `final_state_vector`), global phase included
 *   (every gate's exp(i*pi*t*global_shift) and the e^{i*pi*t/2} of cirq's X**t / Y**t).
 * Observables need not be installed. */
int qhbm_statevector(qhbm_engine* h, const int8_t* d_bits, int U,
                     const float* d_params, void* d_out_states, void* stream);

/* Computational-basis samples of the final states (tfq.layers.Sample as used at
 * qhbmlib/inference/qnn.py:169,177-181,286-291):
 *   d_out_samples [U, n_shots, n_qubits] int8 (device); shot j of state u is drawn from
 *   |<x|C(params)|x_u>|^2 with a counter-based generator keyed by (seed, u, j), so a call is
 *   reproducible and independent of chunking.
 * shift_gate >= 0 adds `shift` to the exponent of that gate of the installed circuit: one
 * program of tfq.differentiators.ParameterShift.get_gradient_circuits (qnn.py:192-199);
 * pass -1, 0.0 for the unshifted circuit. */
int qhbm_sample(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params,
                int n_shots, uint64_t seed, int shift_gate, double shift,
                int8_t* d_out_samples, void* stream);

/* Shot COUNTS of several parameter-shifted programs in one launch set -- what the sampled
 * estimators reduce their shots to (qnn.py:170-226: tfq.layers.SampledExpectation with the
 * ParameterShift differentiator samples 2 programs per gate occurrence; a Pauli-string estimate is a
 * signed sum of the counts, a BitstringEnergy average a weighted one):
 *   d_out_counts [n_programs, U, 2^n_qubits] int32 (device): how many of the n_shots shots of
 *   C_q(params)|x_u> gave outcome x (index = the bitstring read big-endian, as qhbm_statevector);
 *   program q is the installed circuit with `shifts[q]` added to the exponent of gate
 *   `shift_gates[q]` (HOST arrays of n_programs entries; gate < 0 = the unshifted circuit).
 * Shot j of (q, u) is drawn with a counter-based generator keyed by (seed, q, u, j): a call is
 * reproducible and independent of how the engine cuts the (program, state) pairs into launch sets.
 * n_qubits <= 24 (one counter per outcome); larger registers use qhbm_sample per program. */
int qhbm_sample_counts(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params,
                       int n_programs, const int32_t* shift_gates, const float* shifts,
                       int n_shots, uint64_t seed, int32_t* d_out_counts, void* stream);

/* ---- EBM side -------------------------------------------- */
/* Spin-parity energies of bitstrings on the current HIP device (no engine handle):
 *   d_energy[i] = sum_k d_thetas[k] * prod_{q in S_k} (1 - 2 x_i[q]),  S_k = set bits of d_masks[k]
 * over the COLUMNS of d_bits [n_rows, n_bits] int8.  This is BernoulliEnergy / KOBE
 * (qhbmlib/models/energy.py:123-209: SpinsFromBitstrings -> Parity -> VariableDot,
 * energy_utils.py:39-110) as one kernel; qhbmlib/inference/ebm.py:458-461 evaluates it over
 * all 2^n bitstrings after every variable update. */
int qhbm_parity_energy(const int8_t* d_bits, int64_t n_rows, int n_bits,
                       const uint64_t* d_masks, const float* d_thetas, int n_terms,
                       float* d_energy, void* stream);
/* Its VJP with respect to thetas: d_grad[k] = sum_i d_weights[i] * parity_k(x_i) (overwritten). */
int qhbm_parity_energy_vjp(const int8_t* d_bits, int64_t n_rows, int n_bits,
                           const uint64_t* d_masks, int n_terms, const float* d_weights,
                           float* d_grad, void* stream);

/* ---- introspection (tests, bench, DESIGN.md numbers) ------------------- */
/* Number of HBM passes (kernel launches over the state) the scheduler emits
 * for one forward of the installed circuit + observables. */
int qhbm_num_passes(qhbm_engine* h, int* forward_passes, int* backward_passes);
/* Human-readable description of the schedule, written into buf. */
int qhbm_describe_schedule(qhbm_engine* h, char* buf, size_t buf_len);
/* Accumulated HIP-event time (ms) and launch count of the pass kernels (forward passes, adjoint
 * passes, lambda = O psi) since the last call with reset != 0.  Timing is only recorded when the
 * option "profile_events" is non-zero; it synchronises the stream.  Any out pointer may be NULL. */
int qhbm_kernel_time_ms(qhbm_engine* h, int reset, double* fwd_ms,
                        int64_t* fwd_launches, double* bwd_ms,
                        int64_t* bwd_launches, double* obs_ms,
                        int64_t* obs_launches);
/* HBM bytes one call on U states must move under the installed schedule if every tile a pass
 * touches is read once and written once (tiles the kernels skip as identically zero excluded):
 * summed over the forward passes, lambda = O psi, and the adjoint passes (with_vjp != 0).  This
 * is the byte count bench.py's roofline divides by the measured kernel time. */
int qhbm_traffic_model(qhbm_engine* h, int U, int with_vjp, double* fwd_bytes,
                       double* obs_bytes, double* bwd_bytes);
/* fp32 operations (one FMA = 2) the gate arithmetic of one call on U states executes under the
 * installed schedule, counted from the plan's instance records with the per-micro-op costs of the
 * kernels' packed-fp32 sequences (X**t as three shears: 6 per amplitude forward, 16 in the adjoint
 * incl. lambda and the inner product; phases, FULL tables, boundary phases on the share of waves that
 * run them; tiles and waves the kernels skip as identically zero excluded; wave reductions, address
 * arithmetic and record decoding NOT counted): forward passes, lambda = O psi, adjoint passes.
 * bench.py divides it by the measured kernel time for a compute roofline against the fp32 vector
 * peak (157.3 TFLOP/s). */
int qhbm_flop_model(qhbm_engine* h, int U, int with_vjp, double* fwd_flops,
                    double* obs_flops, double* bwd_flops);

/* Executed micro-ops of every pass of the forward (adjoint == 0) or backward schedule, in
 * WAVE-EXECUTIONS per state (a micro-op a wave of 64 threads runs once counts 1; tiles and waves the
 * kernels skip excluded): out[pass * QHBM_CENSUS_COLUMNS + column], at most max_passes rows;
 * *n_passes = passes of the schedule.  scripts/instruction_mix.py weighs the per-micro-op
 * instruction counts of the compiled pass kernels with it (the dynamic instruction mix). */
enum {
  QHBM_CENSUS_TILES = 0,         /* workgroups per state */
  QHBM_CENSUS_ROUNDS,            /* register rounds (one LDS exchange each but the first) */
  QHBM_CENSUS_ROUNDS_BARRIER,    /* ... whose exchange needs the barriers */
  QHBM_CENSUS_ROUNDS_NO_BARRIER,
  QHBM_CENSUS_INSTANCES,         /* instance records decoded */
  QHBM_CENSUS_X,                 /* X**t (adjoint: with a gradient slot) */
  QHBM_CENSUS_X_NO_SLOT,         /* adjoint: X**t of a frozen parameter (no inner product) */
  QHBM_CENSUS_FULL,              /* 15-entry diagonal table */
  QHBM_CENSUS_PH1,
  QHBM_CENSUS_PH2,
  QHBM_CENSUS_CPH_TILE_ON,       /* boundary phase, predicate = tile bit, on */
  QHBM_CENSUS_CPH_WAVE_ON,       /* predicate = thread bit that is uniform over the wave, on */
  QHBM_CENSUS_CPH_LANE,          /* predicate varies inside the wave */
  QHBM_CENSUS_CPH_OFF,           /* predicate evaluated, off in the whole wave */
  QHBM_CENSUS_REDUCE8,           /* adjoint: eight-wide wave reductions of gradient partials */
  QHBM_CENSUS_COLUMNS
};
int qhbm_op_census(qhbm_engine* h, int adjoint, int max_passes, double* out, int* n_passes);

/* Plan searches this engine has run so far: forward plans (rebuilt when the circuit, the observables or a planning
 * option changes) and backward plans (also per gradient mask; a mask the engine has seen before is served from its
 * cache and builds nothing).  Either pointer may be NULL. */
int qhbm_plan_builds(qhbm_engine* h, int64_t* forward_plans, int64_t* backward_plans);
/* Sustained packed-fp32 rate and shader clock of the device RIGHT NOW: one probe launch (every SIMD issues
 * v_pk_fma_f32 from four waves, the pass kernels' occupancy) timed with HIP events, s_memtime / s_memrealtime read
 * inside the waves.  ghz: shader clock during the probe; cycles_per_pk_fma: issue cost per wave instruction and SIMD;
 * tflops: the packed-fp32 rate that follows (4 flop x 64 lanes per instruction) -- the ATTAINABLE ceiling bench.py
 * prints beside the nominal 157.3 TFLOP/s.  Synchronises `stream`.  Any out pointer may be NULL. */
int qhbm_clock_probe(qhbm_engine* h, double* ghz, double* cycles_per_pk_fma, double* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QHBM_ENGINE_H_ */
