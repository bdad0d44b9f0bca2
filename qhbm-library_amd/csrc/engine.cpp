// engine.cpp -- C ABI (include/qhbm_engine.h) on top of the scheduler and the
// gfx950 kernels.  This is the boundary a QuantumInference subclass binds
// (/root/reference/qhbmlib/inference/qnn.py:82-84,114-139).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iterator>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/qhbm_engine.h"
#include "kernels.h"
#include "program.h"
#include "schedule.h"

using namespace qhbm;

namespace {

std::string g_create_error;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  // Growing an existing buffer adds an eighth on top (per-batch buffers follow the number of unique
  // bitstrings, which wanders from step to step); the first allocation is exact.
  hipError_t reserve(size_t count, bool slack = true) {
    if (count <= n) return hipSuccess;
    if (n && slack) count += count / 8;
    release();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T>& h) {
    hipError_t e = reserve(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  }
};

template <typename T> void swap_buf(DevBuf<T>& a, DevBuf<T>& b) { std::swap(a.p, b.p); std::swap(a.n, b.n); }

struct DevicePlan {
  Plan plan;
  std::vector<PassArgs> args;
  DevBuf<uint32_t> prog, tables, rec_offsets;
  DevBuf<CoefJob> jobs;
  DevBuf<float> coef;
  bool uploaded = false;
  void swap(DevicePlan& o) {
    std::swap(plan, o.plan);
    std::swap(args, o.args);
    swap_buf(prog, o.prog); swap_buf(tables, o.tables); swap_buf(rec_offsets, o.rec_offsets);
    swap_buf(jobs, o.jobs); swap_buf(coef, o.coef);
    std::swap(uploaded, o.uploaded);
  }
};

struct TimedEvent {
  hipEvent_t a, b;
  int kind;  // 0 forward pass, 1 adjoint pass, 2 lambda = O psi
};

}  // namespace

struct qhbm_engine {
  int device = -1;
  std::string err;
  Model model;
  bool have_circuit = false;
  // options
  int opt_tile = 0, opt_adj_tile = 0, opt_profile = 0, opt_round = 0;
  int opt_full_fwd = 60, opt_full_adj = 60, opt_force_general = 0;
  int opt_meas_tile = 0;     // tile qubits of measurement-only passes (0 = largest)
  int opt_values_from_obs = 1;  // single observable: <psi|O|psi> from lambda = O psi, no measurement in the forward sweep
  bool retained_mu = false;     // the retained batch also holds the unweighted lambda = O psi
  int opt_cph_wave_bits = 1; // boundary controlled-phase predicates on wave bits (schedule.h Plan::cph_wave_bits)
  int opt_fwd_values_obs = -1; // forward-only calls, one observable: values from the lambda = O psi kernel (no lambda stored) instead of
                               // measurements in the passes (-1: when the plan has more than one pass, i.e. the state is in HBM anyway)
  int opt_adj_stop_early = -1; // frozen parameters: the backward sweep stops at the first live gate (-1: if the time model says that is cheaper)
  int opt_adj_plan_search = 1; // adjoint: build the scheduler's best few pass orders and keep the one with the fewest model flops
  int opt_wide_last = -1;      // forward: the last gate pass may take a tile one or two bits wider (-1: unless tile_qubits is set)
  int opt_fwd_pair = 1;        // dense lean forward passes run on pairs of states, tiles in registers (pass_fwd2_kernel)
  int opt_shift_prefix = 1;    // parameter shift: a shifted program starts at the pass that holds its gate, from the base program's state
  int opt_adj_relabel = 1;     // adjoint plans move finished index bits out of the 128-byte lines (schedule.h Pass)
  int opt_obs_xcd_states = -1; // lambda = O psi: one state per XCD at a time (1), every XCD an eighth of each state (0); -1: by state size
  int opt_adj_exchange = 1;  // lean adjoint passes: register-resident tile pair + one LDS exchange buffer
  int retained_U = 0;  // final states of the last qhbm_expectation_retain still sit in psi
  int state_grad_U = 0;  // rows of state_grad the last adjoint VJP filled (qhbm_state_gradients)
  int64_t opt_chunk = 0;
  int64_t opt_budget_mb = 0;  // 0: a third of the device's memory, resolved at first use (budget_bytes)
  size_t resolved_budget = 0; // the default budget of THIS engine once a device query has succeeded
  // plans
  bool plans_valid = false;
  bool adj_valid = false;  // the backward plan matches the gradient mask (qhbm_set_gradient_mask invalidates only this)
  // Backward plans (and their device copies) of the gradient masks this model has been used with: two inference paths
  // that alternate on one engine -- the same total circuit once with the data half frozen, once fully trainable --
  // swap plans instead of re-running the plan search and re-uploading on every step.  Keyed by the
  // frozen-parameter vector; cleared whenever the circuit, the observables or a planning option changes.
  std::vector<std::pair<std::vector<char>, std::unique_ptr<DevicePlan>>> adj_cache;
  int64_t fwd_plans_built = 0, adj_plans_built = 0;  // plan searches run so far (qhbm_plan_builds: a cache hit builds nothing)
  bool model_uploaded = false;  // everything upload_model copies to the device is current
  DevicePlan fwd, adj;
  DevBuf<DevTerm> terms, global_terms;  // global_terms: measured on the final state in HBM (too wide for a tile)
  DevBuf<ObsGroup> obs_groups;
  bool terms_by_op = false;  // the uploaded terms / groups are in the (mask, observable) order of gather_multi_mode
  DevBuf<ObsBTerm> obs_bterms;    // the same terms sorted and cut for the block-grouped kernels (observable.hip)
  DevBuf<ObsBGroup> obs_bgroups;
  uint32_t n_obs_bgroups = 0;
  int opt_obs_block_bits = kObsBlockBits;  // shape of the block-grouped kernels (kernels.h): 13 or 12
  int opt_obs_split_rows = 0;              // blocks of 2^13: the halves of the workgroup split the block's rows, not a group's masks
  int opt_obs_kernel = -1;      // lambda = O psi / values: 0 = one gather per mask (apply_observable_kernel), 1 = partner blocks through
                                // LDS (observable_blocks_kernel), -1 = whichever the fitted cost model prices lower (block_kernel())
  mutable int block_choice = -1;  // cached verdict of block_kernel() (-1: not computed for the installed model / options)
  int opt_gather_multi = 0;     // 2..4 observables: 1 = the gather kernel forms lambda AND carries a value accumulator per observable
                                // (one launch).  Measured slower than the block kernel's value launch + the gather launch for lambda
                                // (config 3 as XX / YY / ZZ sums: 103 against 64.7 ms, profiles/r05_xxz3_gather_multi_ab.txt): off
                                // by default, kept selectable (tests, A/B runs)
  int opt_multi_values = -1;    // several observables: values from the block kernel after lean passes (-1: when some term flips >= 2
                                // qubits or needs a measurement-only pass, at most kMultiValueOps observables)
  DevBuf<uint64_t> probe_out;  // qhbm_clock_probe's buffers
  DevBuf<float> probe_sink;
  DevBuf<float> value_part;  // value mode: one partial of <psi|O|psi> per workgroup of apply_observable_kernel
  uint32_t n_obs_groups = 0;
  uint32_t n_gather_terms = 0;  // terms of the gather kernel's FIRST launch (all of them unless far windows took some)
  // Two-level lambda = O psi (>= 26 qubits, gather kernel, single-observable modes): every mask that leaves a
  // workgroup's block makes it read a partner run from the fabric -- at 28 qubits the 17 single flips of a TFIM cost
  // 12.6 reads of the whole 2-GiB state.  Masks that flip only bits of a seven-bit WINDOW [far_hi, far_hi + 7) (and bits
  // 0..3) go to a FAR launch instead, which works in the index space where the window has changed places with bits
  // [4, 11): there they permute the workgroup's own block (LDS copy, no gather).  A far launch reads psi and reads and
  // writes lambda once (3 state-sized transfers) whatever the number of its masks.
  struct FarWindow {
    DevBuf<DevTerm> terms;
    DevBuf<ObsGroup> groups;
    uint32_t n_terms = 0, n_groups = 0, far_hi = 0;
  };
  FarWindow far[3];
  int n_far = 0;
  int opt_far_windows = 0;   // 0 (default): never -- measured: config 5 (28 qubits, 17 far single flips) 65.0 ms in one launch,
                             // 66.2 in three: the far launches move 3 state-sized transfers each as scattered 128-byte runs at about
                             // half the rate of the streaming first launch; 1: always (any window with a mask); -1: from 26 qubits up,
                             // windows with >= 3 masks
  DevBuf<float2> psi, lam;
  DevBuf<float> state_grad, slot_factor, vals_tmp, upstream_tmp, phase_cs;
  DevBuf<ShiftPhase> shift_phases;  // gates with a cirq global_shift (qhbm_statevector restores their phases)
  int n_shift_phases = 0;
  // parameter-shift batches: per-program coefficient buffers, shift tables, accumulators
  DevBuf<float> coef_batch, shift_vals, shift_weight, vals_batch;
  DevBuf<int> shift_gates, shift_param;
  DevBuf<int> shift_dst;             // execution order -> gate order of the shifted programs (prog_acc index)
  std::vector<uint32_t> shift_group_end;  // programs (execution order) whose shifted gate sits in pass <= i end here
  DevBuf<double> prog_acc;
  size_t coef_batch_programs = 0;  // copies of the forward plan's static words already in coef_batch
  bool shift_ready = false;        // shift tables on the device match the model
  bool shift_tables_from_obs = false;  // ... and were ordered for values taken from the observable kernel (or in the passes)
  uint32_t shift_programs = 0, shift_gate_count = 0;
  DevBuf<float> tile_grad;              // [chunk states * tiles, slots of one adjoint pass]
  DevBuf<unsigned long long> vals64;    // [U, n_ops] fixed-point accumulators of the expectation values
  DevBuf<float> op_scale, op_inv_scale; // per op: 2^(+-shift), see program.h kValueFracBits
  std::vector<float> h_op_scale, h_op_inv_scale;
  DevBuf<double> block_cum;
  DevBuf<int> param_slot_begin, param_slots;
  std::vector<TimedEvent> events;
  std::vector<TimedEvent> free_events;
};

namespace {

int fail(qhbm_engine* h, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return 1;
}

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return fail(h, std::string(#expr) + ": " + hipGetErrorString(_e));               \
  } while (0)

template <typename T> size_t buf_bytes(const DevBuf<T>& b) { return b.n * sizeof(T); }
size_t plan_bytes(const DevicePlan& d) {
  return buf_bytes(d.prog) + buf_bytes(d.tables) + buf_bytes(d.rec_offsets) + buf_bytes(d.jobs) + buf_bytes(d.coef);
}
// Every byte of device memory the engine holds (qhbm_allocated_bytes; a host-side engine cache bounds
// its footprint with it).
size_t own_bytes(const qhbm_engine* h) {
  return buf_bytes(h->psi) + buf_bytes(h->lam) + buf_bytes(h->state_grad) + buf_bytes(h->tile_grad) +
         buf_bytes(h->vals64) + buf_bytes(h->block_cum) + buf_bytes(h->coef_batch) + buf_bytes(h->vals_batch) +
         buf_bytes(h->value_part) + buf_bytes(h->probe_out) + buf_bytes(h->probe_sink) + buf_bytes(h->upstream_tmp) + buf_bytes(h->vals_tmp) + buf_bytes(h->prog_acc) +
         buf_bytes(h->shift_vals) + buf_bytes(h->shift_weight) + buf_bytes(h->shift_gates) + buf_bytes(h->shift_param) + buf_bytes(h->shift_dst) +
         buf_bytes(h->slot_factor) + buf_bytes(h->phase_cs) + buf_bytes(h->shift_phases) + buf_bytes(h->terms) +
         buf_bytes(h->global_terms) + buf_bytes(h->obs_groups) + buf_bytes(h->far[0].terms) + buf_bytes(h->far[0].groups) +
         buf_bytes(h->far[1].terms) + buf_bytes(h->far[1].groups) + buf_bytes(h->far[2].terms) + buf_bytes(h->far[2].groups) + buf_bytes(h->obs_bterms) + buf_bytes(h->obs_bgroups) + buf_bytes(h->op_scale) + buf_bytes(h->op_inv_scale) +
         buf_bytes(h->param_slot_begin) + buf_bytes(h->param_slots) + plan_bytes(h->fwd) + plan_bytes(h->adj) + [&] {
           size_t cached = 0;  // backward plans of other gradient masks, kept with their device copies (adj_cache)
           for (const auto& kv : h->adj_cache) cached += plan_bytes(*kv.second);
           return cached;
         }();
}

int need_device(qhbm_engine* h) {
  if (h->device < 0)
    return fail(h, "engine was created without a device (planning only); no CPU fallback exists");
  HIPCHK(hipSetDevice(h->device));
  return 0;
}

void fill_args(const Plan& plan, const Model& m, std::vector<PassArgs>* args, std::vector<uint32_t>* prog,
               std::vector<uint32_t>* tables) {
  args->clear();
  prog->clear();
  tables->clear();
  for (const Pass& p : plan.passes) {
    PassArgs a;
    std::memset(&a, 0, sizeof(a));
    a.flags = p.flags;
    a.n = uint32_t(plan.n_eff);
    a.c = uint32_t(p.c);
    a.n_nonlocal = uint32_t(p.nonlocal_pos.size());
    a.prog_off = uint32_t(prog->size());
    a.spread_off = uint32_t(tables->size());
    a.spread_shift = 0;  // K == c: the only entry is 0
    for (size_t i = size_t(p.c); i < p.local_phys.size(); ++i) {
      if (i == size_t(p.c)) a.spread_shift = uint32_t(p.local_phys[i]);
      else if (p.local_phys[i] != p.local_phys[i - 1] + 1) { a.spread_shift = 0xffffffffu; break; }
    }
    a.n_ops = uint32_t(m.n_ops);
    a.slot_base = uint32_t(p.slot_base);
    a.n_slots = uint32_t(p.n_slots);
    for (size_t i = 0; i < p.nonlocal_phys.size(); ++i) a.nonlocal_pos[i] = uint8_t(p.nonlocal_phys[i]);
    for (size_t i = 0; i < p.local_phys.size(); ++i) a.local_pos[i] = uint8_t(p.local_phys[i]);
    for (size_t bit = 0; bit < 32; ++bit) a.phys_of[bit] = uint8_t(bit < p.phys_of.size() ? p.phys_of[bit] : int(bit));
    for (size_t bit = 0; bit < 32; ++bit) a.log_of[a.phys_of[bit]] = uint8_t(bit);
    for (size_t i = 0; i < p.nonlocal_phys.size(); ++i) a.nonlocal_mask |= 1u << p.nonlocal_phys[i];
    for (uint32_t I = 0; I < 8; ++I) {  // (the pass kernels give a thread eight float4 of its tile: kernels.hip prefetch_tile)
      const uint32_t l = p.K >= 3 ? I << (p.K - 3) : 0u, hi = l >> a.c;
      a.row_off[I] = (l & ((1u << a.c) - 1u)) |
                     (a.spread_shift != 0xffffffffu ? hi << a.spread_shift : (hi < p.spread.size() ? p.spread[hi] : 0u));
    }
    a.frozen_old_local = p.frozen_old_local;
    prog->insert(prog->end(), p.prog.begin(), p.prog.end());
    tables->insert(tables->end(), p.spread.begin(), p.spread.end());
    a.tl_off = uint32_t(tables->size());
    tables->insert(tables->end(), p.round_tl.begin(), p.round_tl.end());
    if (p.flags & PASS_RELABEL) {
      a.frozen_new_local = p.frozen_new_local;
      while (tables->size() % 4) tables->push_back(0u);  // the store reads (l0, off0, l1, off1) as one 16-byte word
      a.relabel_off = uint32_t(tables->size());
      tables->insert(tables->end(), p.relabel_tab.begin(), p.relabel_tab.end());
      // consecutive live out-indices are adjacent amplitudes iff the lowest position the tile owns is address bit 0
      a.relabel_pairs = (p.local_phys[0] == 0 && !(p.frozen_new_local == ((1u << p.K) - 1u))) ? 1u : 0u;
      uint32_t n_live = 0;
      for (int i = 0; i < p.K; ++i) n_live += !(p.frozen_new_local >> i & 1u);
      if (n_live < 1) a.relabel_pairs = 0;
      std::memset(a.fz_src, 0xff, sizeof(a.fz_src));
      for (int i = 0; i < p.K; ++i)
        if (p.frozen_new_local >> i & 1u) {
          a.fz_local_bit[a.n_fz] = uint8_t(i);
          a.fz_out_pos[a.n_fz] = uint8_t(p.store_local_phys[size_t(i)]);
          a.fz_src[p.store_local_phys[size_t(i)]] = uint8_t(i);
          ++a.n_fz;
        }
      // (kernels.hip store_tile_relabeled: a thread's out-indices are (tid + i * threads) [* 2 for pairs])
      const uint32_t threads = 1u << (p.K - 4), entries = uint32_t(p.relabel_tab.size() / 2);
      const uint32_t count = a.relabel_pairs ? entries / 2 : entries;
      uint32_t shift = uint32_t(p.K - 4) + (a.relabel_pairs ? 1u : 0u);
      a.relabel_iters = (count + threads - 1) / threads;
      a.relabel_l1 = entries > 1 ? p.relabel_tab[2] : 0u;
      for (uint32_t i = 0; i < 16; ++i) {
        const uint64_t o = uint64_t(i) << shift;
        a.relabel_hi[i][0] = o < entries ? p.relabel_tab[2 * size_t(o)] : 0u;
        a.relabel_hi[i][1] = o < entries ? p.relabel_tab[2 * size_t(o) + 1] : 0u;
      }
    }
    args->push_back(a);
  }
  if (!plan.adjoint) {
    // Amplitude-space pruning of the head of the forward sweep: an index bit no non-diagonal op has
    // acted on yet still equals the input bitstring wherever psi is non-zero.  A tile of pass p that
    // differs from the input on such a bit among its NON-LOCAL bits is identically zero, stays zero
    // under the pass, and already reads as zeros in HBM (the first pass wrote them): the kernel
    // returns at once.  Config 3: half the tiles of the second (and heaviest) forward pass.
    uint32_t touched = 0;  // (forward plans never relabel: logical = physical positions)
    for (size_t i = 0; i < args->size(); ++i) {
      uint32_t nl = 0;
      for (uint32_t k = 0; k < (*args)[i].n_nonlocal; ++k) nl |= 1u << (*args)[i].nonlocal_pos[k];
      if (i > 0) (*args)[i].zero_mask = nl & ~touched;
      touched |= plan.passes[i].mat_bits;
    }
  }
  if (plan.adjoint) {
    // Amplitude-space pruning of the tail of the backward sweep.  At the start of adjoint pass p,
    // psi is (the circuit's FIRST ops, those of passes p..last) applied to the basis state, and no
    // bit that is non-local in all of those passes has been acted on other than diagonally: psi is
    // zero wherever such a bit differs from the input bitstring, so those tiles add nothing to any
    // gradient, and lambda there is only ever paired with zero psi in later passes.  The kernel
    // skips them outright.
    // "Acted on" means by a non-diagonal op of pass p or later; a bit that is local in a later pass
    // without a gate there (the low bits every tile holds, padding) is as good: the tiles skipped now
    // are read again then, but hold zeros of psi (to rounding) next to a stale lambda.
    uint32_t later_mat = plan.dense_tail ? ~0u : 0u;  // logical bits (dense_tail: the sweep does not end at the basis state)
    for (size_t i = args->size(); i-- > 0;) {
      uint32_t nl = 0;
      for (uint32_t k = 0; k < (*args)[i].n_nonlocal; ++k) nl |= 1u << (*args)[i].nonlocal_pos[k];
      later_mat |= plan.passes[i].mat_bits;
      uint32_t later_phys = 0;  // ... at the positions they have when pass i loads its tiles
      for (size_t bit = 0; bit < plan.passes[i].phys_of.size(); ++bit)
        if (later_mat >> bit & 1u) later_phys |= 1u << plan.passes[i].phys_of[bit];
      (*args)[i].zero_mask = nl & ~later_phys;
    }
  }
  for (size_t i = 0; i < args->size(); ++i) {
    PassArgs& a = (*args)[i];
    // the first forward pass writes ONE tile per state when nothing has to be zero-filled: every tile-id bit
    // follows the input bitstring
    if ((a.flags & PASS_INIT_BASIS) && (a.flags & PASS_NO_ZERO_FILL))
      for (uint32_t k = 0; k < a.n_nonlocal; ++k) a.zero_mask |= 1u << a.nonlocal_pos[k];
    a.n_free = a.n_nonlocal - uint32_t(__builtin_popcount(a.zero_mask));
  }
}

double pass_flops_per_amplitude(const Plan& plan, const Pass& p);

// Modelled time of the adjoint sweep per state, in seconds: per pass the larger of its arithmetic at the rate the
// pass kernel sustains (67 TFLOP/s of qhbm_flop_model's flops: 65-69 over 20...28 qubits, depths 16 and 32, tiles of
// 2^12 and 2^13, scripts/adj_tile_ab.sh) and its tile traffic (qhbm_traffic_model's bytes) at 5.5 TB/s, the rate of
// a pass with nothing to compute (config 3: 24.6 ms per 4096 state pairs read and written).
double adjoint_plan_seconds(const Plan& plan, const Model& m) {
  std::vector<PassArgs> args;
  std::vector<uint32_t> prog, tables;
  fill_args(plan, m, &args, &prog, &tables);
  const double amps = double(size_t(1) << plan.n_eff);
  double t = 0.0;
  for (size_t i = 0; i < args.size(); ++i) {
    const Pass& p = plan.passes[i];
    const double live = 1.0 / double(1ull << __builtin_popcount(args[i].zero_mask));
    int low_missing = 0;  // a tile without some of the four low index bits still moves whole 128-byte lines
    for (uint32_t k = 0; k < args[i].n_nonlocal; ++k) low_missing += args[i].nonlocal_pos[k] < 4;
    const double lines = std::min(1.0, live * double(1u << low_missing));
    const double write_share = (p.flags & PASS_RELABEL) ? 1.0 / double(1u << __builtin_popcount(p.frozen_new_local)) : 1.0;
    const double bytes = lines * amps * 8.0 * 2.0 * (1.0 + ((p.flags & PASS_STORE) ? write_share : 0.0));
    const double flops = live * amps * pass_flops_per_amplitude(plan, p);
    // (+ a tenth of the traffic time: what the start and the end of a pass -- tiles loading, tiles draining -- add to
    // a pass bound by arithmetic; fitted on orders with 6...8 passes more than their rivals, scripts/adj_search_ab.sh)
    t += std::max(flops / 67e12, bytes / 5.5e12) + 0.1 * bytes / 5.5e12;
  }
  return t;
}

int build_plans(qhbm_engine* h) {
  if (h->plans_valid && h->adj_valid) return 0;
  h->retained_U = 0;
  if (!h->have_circuit) return fail(h, "qhbm_set_circuit has not been called");
  if (h->model.n_ops > kMaxOps) return fail(h, "too many observables (max 1024)");
  std::string err;
  if (!h->plans_valid) {  // (a change of the gradient mask alone keeps the forward plan: it does not depend on it)
    h->adj_cache.clear();
    if (!build_plan(h->model, h->opt_tile, h->opt_round, false, &h->fwd.plan, &err, h->opt_full_fwd, h->opt_meas_tile,
                    h->opt_cph_wave_bits != 0, false, h->opt_wide_last))
      return fail(h, "forward plan: " + err);
    h->fwd.uploaded = false;
    ++h->fwd_plans_built;
  }
  // The backward plan.  Its pass kernel runs at the same fp32 rate whatever the plan (adjoint_plan_seconds), so the
  // plan with the least modelled time -- arithmetic, or tile traffic where a pass has little to compute -- is kept:
  //  * the scheduler's search ranks pass orders by a proxy (gates x live share); the best few complete orders and
  //    the greedy one are built out and compared;
  //  * with the tile size left to the engine, the same is done with tiles of 2^13 amplitudes, which win where no
  //    index bit finishes early and every pass is a full pass (depth-32 TFIM: - 5...9 %), and lose on chains that
  //    prune early (config 3: + 8 %).
  const bool relabel = h->opt_adj_relabel != 0 && h->opt_adj_exchange != 0;
  auto plan_adjoint_of = [&](const Model& model, int tile, Plan* out, double* seconds, std::string* e) {
    if (!build_plan(model, tile, 0, true, out, e, h->opt_full_adj, 0, h->opt_cph_wave_bits != 0, relabel)) return false;
    *seconds = adjoint_plan_seconds(*out, model);
    if (!h->opt_adj_plan_search) return true;
    const std::vector<std::vector<uint32_t>> orders = out->candidate_orders;
    for (const std::vector<uint32_t>& order : orders) {
      Plan alt;
      std::string e2;
      if (!build_plan(model, tile, 0, true, &alt, &e2, h->opt_full_adj, 0, h->opt_cph_wave_bits != 0, relabel, -1, &order))
        continue;
      const double f = adjoint_plan_seconds(alt, model);
      if (std::getenv("QHBM_PLAN_DEBUG"))
        std::fprintf(stderr, "[plan] tile %d candidate with %zu passes: %.4g ms per state (kept so far: %zu passes, %.4g ms)\n", tile,
                     alt.passes.size(), f * 1e3, out->passes.size(), *seconds * 1e3);
      if (f < 0.99 * *seconds) { *out = std::move(alt); *seconds = f; }
    }
    return true;
  };
  //  * with frozen parameters (qhbm_set_gradient_mask) the sweep may stop at the first gate of a live parameter --
  //    but what is left of psi there is no basis state, and nothing can be pruned any more: for a shallow circuit
  //    the full sweep with its pruned tail is the cheaper one.  Both are planned.
  auto plan_adjoint = [&](int tile, Plan* out, double* seconds, std::string* e) {
    bool any_live = false;  // (no live parameter at all: nothing to sweep, whatever the option says)
    for (const Gate& G : h->model.gates) any_live |= G.param_idx >= 0 && !h->model.frozen(G.param_idx);
    if (h->opt_adj_stop_early == 0 && !h->model.param_frozen.empty() && any_live) {
      Model whole = h->model;
      whole.stop_at_first_live_gate = false;
      return plan_adjoint_of(whole, tile, out, seconds, e);
    }
    if (!plan_adjoint_of(h->model, tile, out, seconds, e)) return false;
    if (h->opt_adj_stop_early < 0 && !h->model.param_frozen.empty() && out->dense_tail) {
      Model whole = h->model;
      whole.stop_at_first_live_gate = false;
      Plan alt;
      std::string e2;
      double f = 0.0;
      if (plan_adjoint_of(whole, tile, &alt, &f, &e2) && f < *seconds) { *out = std::move(alt); *seconds = f; }
    }
    return true;
  };
  double adj_seconds = 0.0;
  if (!plan_adjoint(h->opt_adj_tile, &h->adj.plan, &adj_seconds, &err)) return fail(h, "adjoint plan: " + err);
  if (h->opt_adj_tile == 0 && h->opt_adj_exchange != 0 && h->adj.plan.K == 12 && h->adj.plan.passes.size() > 1 &&
      h->adj.plan.n_eff >= 14) {
    Plan wide;
    std::string err2;
    double wide_seconds = 0.0;
    // (tiles of 2^13 run 3 - 4 % slower against 2^12 than the model says -- two workgroups of eight waves per CU overlap
    // their tile I/O worse than four of four; seven shapes, profiles/r05_adjoint_tile_choice.txt: they must win by that)
    if (plan_adjoint(13, &wide, &wide_seconds, &err2) && wide.K == 13 && wide_seconds < 0.966 * adj_seconds)
      h->adj.plan = std::move(wide);
  }
  h->adj.uploaded = false;
  ++h->adj_plans_built;
  h->model_uploaded = false;
  h->shift_ready = false;
  if (!h->plans_valid) h->coef_batch_programs = 0;
  h->plans_valid = true;
  h->adj_valid = true;
  return 0;
}

int upload_plan(qhbm_engine* h, DevicePlan* d) {
  if (d->uploaded) return 0;
  std::vector<uint32_t> prog, tables;
  fill_args(d->plan, h->model, &d->args, &prog, &tables);
  HIPCHK(d->prog.upload(prog));
  HIPCHK(d->tables.upload(tables));
  HIPCHK(d->jobs.upload(d->plan.jobs));
  HIPCHK(d->rec_offsets.upload(d->plan.record_offsets));
  {
    std::vector<float> init(d->plan.coef_init.size() + 64, 0.f);
    std::memcpy(init.data(), d->plan.coef_init.data(), d->plan.coef_init.size() * sizeof(uint32_t));
    HIPCHK(d->coef.upload(init));
  }
  d->uploaded = true;
  return 0;
}

// One term of the block-grouped kernels (kernels.h ObsBTerm): slot bits of the block layout are index bits 0, 10, 11, 12
// (blocks of 2^12: 0, 10, 11).
ObsBTerm obs_block_term(const DevTerm& d, bool new_mask, int bb) {
  const uint32_t xin = d.x & ((1u << bb) - 1u), rows = (1u << (bb - 10)) - 1u;
  const uint32_t zs = (d.z & 1u) | (((d.z >> 10) & rows) << 1);
  // variant (scripts/gen_observable_asm.py): Z bits of the three low slot bits | base sign << 3 | odd x << 4 | imaginary << 5
  const uint32_t variant = (zs & 7u) | ((xin & 1u) << 4) | ((d.ny & 1u) << 5);
  const uint32_t chunk = kObsChunkBytes, preamble = kObsPreambleBytes;
  ObsBTerm t;
  t.coeff = d.coeff;
  t.zt = (d.z >> 1) & 511u;
  t.zb = d.z >> bb;
  t.xrow = (((xin >> 1) & 511u) << 4) | (((xin >> 10) & rows) << 13);
  t.off0 = variant * chunk + preamble;
  t.off1 = (variant | (zs & 8u)) * chunk + preamble;
  // sign at the own index: (-1)^ny from the x & z overlap, and i^2 = -1 once ny >= 2
  t.meta = (d.op & 1023u) | (new_mask ? kObsNewMask : 0u) | (((d.ny + (d.ny >> 1)) & 1u) ? kObsSignBit : 0u);
  t.pad = 0u;
  return t;
}

// Copies plans, observable tables and the parameter -> slot map to the device ONCE per model: a
// compute call on an unchanged model issues no host copy and no synchronisation (it can be captured
// into a hipGraph by the caller).
bool gather_multi_mode(const qhbm_engine* h);  // (below, with the other mode predicates)

int upload_model(qhbm_engine* h) {
  if (int rc = build_plans(h)) return rc;
  if (h->terms.p && h->terms_by_op != gather_multi_mode(h)) {  // an option changed which order the gather kernel wants
    h->terms.release();
    h->model_uploaded = false;
  }
  if (h->model_uploaded) return 0;
  if (int rc = upload_plan(h, &h->fwd)) return rc;
  if (int rc = upload_plan(h, &h->adj)) return rc;
  if (!h->terms.p) {
    // lambda = O psi gathers psi[j ^ x] once per distinct x mask: sort by x, cut into groups
    std::vector<DevTerm> t;
    for (const PauliTerm& pt : h->model.terms)
      t.push_back(DevTerm{pt.coeff, pt.x, pt.z, uint32_t(pt.ny), uint32_t(pt.op)});
    // (several observables with a value accumulator each -- gather_multi_mode --: by mask, then by observable; a group is
    // one observable's share of a mask and re-uses the partners its predecessor gathered when the mask is the same)
    const bool by_op = gather_multi_mode(h);
    h->terms_by_op = by_op;
    // the gather kernel's tables of a term list: sorted by mask (then observable), cut into groups, sign classes in order
    auto gather_tables = [&](std::vector<DevTerm>& t, std::vector<ObsGroup>* groups) {
      std::stable_sort(t.begin(), t.end(), [by_op](const DevTerm& a, const DevTerm& b) {
        return a.x != b.x ? a.x < b.x : (by_op && a.op < b.op);
      });
      groups->clear();
      for (size_t k = 0; k < t.size(); ++k) {
        const bool new_mask = k == 0 || t[k].x != t[k - 1].x;
        if (new_mask || (by_op && t[k].op != t[k - 1].op) || k % (kObsTermChunk / 2) == 0) {
          // same_x bit 0 ("re-use the predecessor's partners") only in (mask, observable) order, where bit 1 makes every
          // group fetch ALL its pairs: in the default order a mask cut at a chunk boundary gathers again, because its first
          // part may have fetched only the pairs it does not vanish on (or nothing at all)
          groups->push_back(ObsGroup{t[k].x, 0, 0, 0, 0, t[k].op, by_op ? (new_mask ? 2u : 3u) : 0u});
        }
        groups->back().end = uint32_t(k + 1);
        groups->back().has_imag |= t[k].ny & 1u;
      }
      // order the terms of every real-weight group by sign class (kernels.h ObsGroup)
      const uint32_t smask = obs_slot_mask(uint32_t(h->fwd.plan.n_eff));
      auto cls = [&](const DevTerm& d) { return (d.z & kObsThreadMask) == 0 ? 0 : ((d.z & smask) == 0 ? 1 : 2); };
      size_t begin = 0;
      for (ObsGroup& g : *groups) {
        if (!g.has_imag) {
          std::stable_sort(t.begin() + begin, t.begin() + g.end,
                           [&](const DevTerm& a, const DevTerm& b) { return cls(a) < cls(b); });
          for (size_t k = begin; k < g.end; ++k) {
            g.n_h += cls(t[k]) == 0;
            g.n_l += cls(t[k]) == 1;
          }
        }
        begin = g.end;
      }
    };
    // (the kernel skips a group's gather on same_x bit 0 and trusts that the predecessor fetched EVERY pair, which only bit 1
    // guarantees: round 5's wrong-value bug was a table with bit 0 alone)
    auto tables_ok = [](const std::vector<ObsGroup>& groups) {
      for (const ObsGroup& g : groups)
        if ((g.same_x & 1u) && !(g.same_x & 2u)) return false;
      return true;
    };
    std::vector<DevTerm> full = t;  // (every term, in the gather order: what the block-grouped tables below start from)
    {
      std::vector<ObsGroup> unused;
      gather_tables(full, &unused);
    }
    // far windows of the two-level sweep (qhbm_engine::FarWindow): top seven bits first
    h->n_far = 0;
    const int n_eff = h->fwd.plan.n_eff;
    const bool far_on = !by_op && obs_amps_per_thread(uint32_t(n_eff)) == 8u &&
                        (h->opt_far_windows > 0 || (h->opt_far_windows < 0 && n_eff >= 26));
    if (far_on) {
      for (int hi = n_eff - 7; hi >= 11 && h->n_far < 3; hi -= 7) {
        const uint32_t window = 0x7fu << hi;
        auto takes = [&](const DevTerm& d) { return (d.x & window) != 0u && (d.x & ~(window | 0xfu)) == 0u; };
        std::vector<uint32_t> masks;
        for (const DevTerm& d : t) if (takes(d)) masks.push_back(d.x);
        std::sort(masks.begin(), masks.end());
        masks.erase(std::unique(masks.begin(), masks.end()), masks.end());
        if (masks.size() < (h->opt_far_windows > 0 ? 1u : 3u)) continue;  // (three state-sized transfers must buy something)
        auto virt = [&](uint32_t v) {  // physical -> virtual index bits: the window changes places with bits [4, 11)
          const uint32_t lo = (v >> 4) & 0x7fu, top = (v >> hi) & 0x7fu;
          return (v & ~((0x7fu << 4) | window)) | (top << 4) | (lo << hi);
        };
        std::vector<DevTerm> ft, rest;
        for (const DevTerm& d : t) {
          if (takes(d)) ft.push_back(DevTerm{d.coeff, virt(d.x), virt(d.z), d.ny, d.op});
          else rest.push_back(d);
        }
        t.swap(rest);
        std::vector<ObsGroup> fg;
        gather_tables(ft, &fg);
        if (!tables_ok(fg)) return fail(h, "observable tables: a group re-uses partners its predecessor need not have fetched");
        qhbm_engine::FarWindow& w = h->far[h->n_far++];
        HIPCHK(w.terms.upload(ft));
        HIPCHK(w.groups.upload(fg));
        w.n_terms = uint32_t(ft.size());
        w.n_groups = uint32_t(fg.size());
        w.far_hi = uint32_t(hi);
      }
    }
    std::vector<ObsGroup> groups;
    gather_tables(t, &groups);
    if (!tables_ok(groups)) return fail(h, "observable tables: a group re-uses partners its predecessor need not have fetched");
    if (t.empty()) t.push_back(DevTerm{0.f, 0u, 0u, 0u, 0u});  // (every term went to a far window: the first launch still writes lambda = 0)
    HIPCHK(h->terms.upload(t));
    HIPCHK(h->obs_groups.upload(groups));
    h->n_obs_groups = uint32_t(groups.size());
    h->n_gather_terms = uint32_t(groups.empty() ? 0u : groups.back().end);
    {  // block-grouped order (kernels.h ObsBTerm): by partner block x >> block bits, then by mask, then by observable
      const int bb = h->opt_obs_block_bits;
      std::vector<DevTerm> bt = full;
      std::stable_sort(bt.begin(), bt.end(), [bb](const DevTerm& a, const DevTerm& b) {
        const uint32_t ao = a.x >> bb, bo = b.x >> bb;
        if (ao != bo) return ao < bo;
        if (a.x != b.x) return a.x < b.x;
        return a.op < b.op;
      });
      std::vector<ObsBTerm> terms2;
      std::vector<ObsBGroup> groups2;
      // A UNIT = the masks one step of the kernel applies from one fetched partner block.  The phase-timing build of
      // the kernel (-DQHBM_OBS_TIMING, profiles/r04_c4_observable_experiments.txt) shows a step of ~2750 cycles as
      // 15 - 25 % the 64-KiB store burst into LDS, 33 - 43 % the masks (~790 cycles per term and wave: 8 KiB of LDS reads
      // each), 18 - 32 % barrier skew, and NO time waiting for the prefetched blocks -- but the pipeline runs only three
      // blocks ahead, and after a group with many masks (x_out = 0: 65 of config 4's 480) it would.  Inside a window of
      // partner blocks (x_out >> 5: what the L2 holds at a time -- the order inside it is free) heavy and light units
      // therefore ALTERNATE (config 4: 53.6 -> 52.8 ms).  Cutting a heavy group into several units that fetch the block
      // again was measured and lost (units of at most 16 / 8 / 4 / 2 terms: 54.7 / 57.1 / 62.6 / 75.4 ms);
      // QHBM_OBS_UNIT_TERMS keeps the knob.  Dealing a group's odd term to BOTH halves by slot rows was measured too
      // (55.5 against 51.4 ms: scripts/experiments/patches/obs_row_split_terms.patch).
      static const size_t cap = std::getenv("QHBM_OBS_UNIT_TERMS") ? size_t(std::max(1, std::atoi(std::getenv("QHBM_OBS_UNIT_TERMS")))) : ~size_t(0) / 2;
      static const bool interleave = !std::getenv("QHBM_OBS_NO_INTERLEAVE");
      struct Unit { uint32_t xo; std::vector<std::pair<size_t, size_t>> masks; size_t terms = 0; };
      std::vector<Unit> units;
      for (size_t k = 0; k < bt.size();) {
        const uint32_t xo = bt[k].x >> bb;
        size_t e = k;
        while (e < bt.size() && (bt[e].x >> bb) == xo) ++e;
        std::vector<std::pair<size_t, size_t>> masks;  // the masks of the group [k, e) and their term ranges
        for (size_t i = k; i < e;) {
          size_t j = i;
          while (j < e && bt[j].x == bt[i].x) ++j;
          masks.emplace_back(i, j);
          i = j;
        }
        std::stable_sort(masks.begin(), masks.end(), [](const std::pair<size_t, size_t>& a, const std::pair<size_t, size_t>& b) {
          return a.second - a.first > b.second - b.first;
        });
        const size_t first_unit = units.size();
        for (const auto& m : masks) {  // first fit, largest first
          const size_t len = m.second - m.first;
          size_t u = first_unit;
          while (u < units.size() && units[u].terms + len > cap && units[u].terms > 0) ++u;
          if (u == units.size()) { units.emplace_back(); units.back().xo = xo; }
          units[u].masks.push_back(m);
          units[u].terms += len;
        }
        k = e;
      }
      if (interleave) {  // inside every window: the heaviest unit while the running load is behind the average, else the lightest
        const int window = 5 + (kObsBlockBits - bb);  // (2 MiB of partner blocks)
        std::vector<Unit> ordered;
        for (size_t a = 0; a < units.size();) {
          size_t b = a;
          while (b < units.size() && (units[b].xo >> window) == (units[a].xo >> window)) ++b;
          std::vector<Unit> win(std::make_move_iterator(units.begin() + long(a)), std::make_move_iterator(units.begin() + long(b)));
          std::stable_sort(win.begin(), win.end(), [](const Unit& x, const Unit& y) { return x.terms > y.terms; });
          double total = 0.0;
          for (const Unit& u : win) total += double(u.terms);
          const double avg = total / double(win.size());
          size_t lo = 0, hi = win.size();
          double done = 0.0;
          for (size_t step = 0; lo < hi; ++step) {
            const bool heavy = done <= avg * double(step);
            Unit& pick = heavy ? win[lo] : win[hi - 1];
            done += double(pick.terms);
            ordered.push_back(std::move(pick));
            if (heavy) ++lo; else --hi;
          }
          a = b;
        }
        units.swap(ordered);
      }
      for (const Unit& u : units) {
        // the unit's masks dealt to the two half-workgroups by term count, largest first (the halves run in step: a unit
        // costs what its larger half costs)
        std::vector<size_t> order(u.masks.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
          return u.masks[a].second - u.masks[a].first > u.masks[b].second - u.masks[b].first;
        });
        std::vector<size_t> half[2];
        size_t load[2] = {0, 0};
        for (size_t m : order) {
          const int hsel = bb == kObsBlockBits && load[1] < load[0] ? 1 : 0;  // (blocks of 2^12: one workgroup, every mask)
          half[hsel].push_back(m);
          load[hsel] += u.masks[m].second - u.masks[m].first;
        }
        ObsBGroup g{u.xo, uint32_t(terms2.size()), 0u, 0u};
        for (int hsel = 0; hsel < 2; ++hsel) {
          // (mask order = x order: neighbouring masks, neighbouring rows)
          std::sort(half[hsel].begin(), half[hsel].end(), [&](size_t a, size_t b) { return u.masks[a].first < u.masks[b].first; });
          for (size_t m : half[hsel])
            for (size_t i = u.masks[m].first; i < u.masks[m].second; ++i)
              terms2.push_back(obs_block_term(bt[i], i == u.masks[m].first, bb));
          if (hsel == 0) g.mid = uint32_t(terms2.size());
        }
        g.end = uint32_t(terms2.size());
        groups2.push_back(g);
      }
      terms2.push_back(ObsBTerm{0.f, 0u, 0u, 0u, 0u, 0u, 0u, 0u});  // (the kernel reads one record ahead)
      HIPCHK(h->obs_bterms.upload(terms2));
      HIPCHK(h->obs_bgroups.upload(groups2));
      h->n_obs_bgroups = uint32_t(groups2.size());
    }
    HIPCHK(h->op_scale.upload(h->h_op_scale));
    HIPCHK(h->op_inv_scale.upload(h->h_op_inv_scale));
  }
  {
    std::vector<DevTerm> gt;
    for (int ti : h->fwd.plan.global_terms) {
      const PauliTerm& pt = h->model.terms[size_t(ti)];
      gt.push_back(DevTerm{pt.coeff, pt.x, pt.z, uint32_t(pt.ny), uint32_t(pt.op)});
    }
    HIPCHK(h->global_terms.upload(gt));
  }
  // parameter -> slots map for the adjoint reduction
  const Plan& ap = h->adj.plan;
  std::vector<int> begin(size_t(h->model.n_params) + 1, 0), slots(ap.slot_gate.size());
  for (int g : ap.slot_gate) ++begin[size_t(h->model.gates[g].param_idx) + 1];
  for (size_t p = 0; p < size_t(h->model.n_params); ++p) begin[p + 1] += begin[p];
  std::vector<int> cursor(begin.begin(), begin.end() - 1);
  for (size_t s = 0; s < ap.slot_gate.size(); ++s) slots[size_t(cursor[h->model.gates[ap.slot_gate[s]].param_idx]++)] = int(s);
  HIPCHK(h->param_slot_begin.upload(begin));
  HIPCHK(h->param_slots.upload(slots));
  HIPCHK(h->slot_factor.upload(ap.slot_factor));
  {
    std::vector<ShiftPhase> sp;
    for (const Gate& G : h->model.gates)
      if (G.global_shift != 0.f && G.kind != QHBM_GATE_I)
        sp.push_back(ShiftPhase{G.param_idx, G.param_idx >= 0 ? G.scalar : 0.f, G.offset, G.global_shift});
    for (const auto& gp : h->fwd.plan.gate_phases) {  // lowered SWAP / ISWAP powers (schedule.cpp lower())
      const Gate& G = h->model.gates[size_t(gp.first)];
      sp.push_back(ShiftPhase{G.param_idx, G.param_idx >= 0 ? G.scalar : 0.f, G.offset, gp.second});
    }
    if (h->fwd.plan.const_phase != 0.0)  // lowered constant Hadamards (schedule.cpp lower())
      sp.push_back(ShiftPhase{-1, 0.f, 1.f, float(h->fwd.plan.const_phase)});
    HIPCHK(h->shift_phases.upload(sp));
    h->n_shift_phases = int(sp.size());
  }
  h->model_uploaded = true;
  return 0;
}

size_t state_bytes(const qhbm_engine* h) { return size_t(8) << h->fwd.plan.n_eff; }

// Workspace budget in bytes (statevector workspace: psi, or psi + lambda).  Default: a third of the
// device's memory (96 GB of the MI355X's 288 GB, so BASELINE config 3's 4096 states x (psi, lambda) x
// 8 MiB = 64 GiB stay resident in one chunk) -- but never more than 45 % of what is FREE when the engine
// first needs it plus what it already holds, so that a second engine, another inference object or
// another rank sharing the GPU sizes itself to what is left instead of to the whole device.  Resolved
// once per engine (never process-wide: a planning-only engine or a failed query must not leave its
// 16 GiB reporting figure behind for a real engine on the same device index).
size_t budget_bytes(qhbm_engine* h) {
  if (h->opt_budget_mb > 0) return size_t(h->opt_budget_mb) << 20;
  if (h->resolved_budget) return h->resolved_budget;
  size_t free_b = 0, total_b = 0;
  if (h->device < 0 || hipSetDevice(h->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b)
    return size_t(16) << 30;  // planning-only engine / failed query: the figure is only reported, not cached
  const size_t share = size_t(0.45 * double(free_b + own_bytes(h)));
  h->resolved_budget = std::max<size_t>(std::min(total_b / 3, share), size_t(64) << 20);
  return h->resolved_budget;
}

uint32_t chunk_states(qhbm_engine* h, int U) {
  // (a chunk is also a grid dimension of the per-state kernels: at most 65535)
  if (h->opt_chunk > 0) return uint32_t(std::min<int64_t>(std::min<int64_t>(h->opt_chunk, U), 65535));
  const size_t budget = budget_bytes(h);
  const size_t fit = std::max<size_t>(1, budget / state_bytes(h));
  return uint32_t(std::min<size_t>(std::min<size_t>(fit, size_t(U)), 65535));
}

hipEvent_t* timer_begin(qhbm_engine* h, int kind, hipStream_t s) {
  if (!h->opt_profile) return nullptr;
  TimedEvent ev;
  if (!h->free_events.empty()) { ev = h->free_events.back(); h->free_events.pop_back(); }
  else { (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b); }
  ev.kind = kind;
  (void)hipEventRecord(ev.a, s);
  h->events.push_back(ev);
  return &h->events.back().b;
}
void timer_end(hipEvent_t* e, hipStream_t s) { if (e) (void)hipEventRecord(*e, s); }

// Forward passes for one chunk.
// One observable: lambda = O psi gives <psi|O|psi> for free (apply_observable_kernel<A, true> / OBS_LAMBDA_VALUE), so
// the paths that compute lambda anyway (value + VJP, and the retained forward of an autograd caller) skip
// every measurement of the forward sweep.
bool value_mode(const qhbm_engine* h) { return h->opt_values_from_obs != 0 && h->model.n_ops == 1; }

// lambda = O psi and the values through partner blocks staged in LDS (observable.hip) instead of one gather per mask
// (apply_observable_kernel).  The gather kernel pays per mask that leaves its block of 2^11 amplitudes (one L2 gather
// of the state each), the block kernel per GROUP of masks with the same partner block of 2^13 (one staged fetch of the
// state each) plus a little per term: picoseconds per amplitude fitted on config 3 (XXZ, 20 qubits: 9 masks leave,
// 8 groups; 5.8 against 7.5) and config 4 (512 random strings, 24 qubits: 453 leave, 173 groups; 177 against 100):
//   gather  2.3 + 0.385 x (masks leaving its block)        block  0.9 + 0.43 x groups + 0.045 x terms
// ("gather_multi_values" = 1: 2..4 observables go through the gather kernel's per-observable accumulators whatever the
// cost model says -- tests and A/B runs)
bool gather_multi_forced(const qhbm_engine* h) {
  return h->opt_gather_multi > 0 && h->model.n_ops >= 2 && h->model.n_ops <= int(kObsGatherMultiOps) &&
         h->opt_values_from_obs != 0 && h->opt_multi_values != 0;
}
int obs_block_shape(const qhbm_engine* h) {
  return h->opt_obs_block_bits == kObsBlockBits && h->opt_obs_split_rows ? kObsShapeRows : h->opt_obs_block_bits;
}
bool block_kernel(const qhbm_engine* h) {
  if (gather_multi_forced(h)) return false;
  if (h->fwd.plan.n_eff < kObsBlockBits || h->opt_obs_kernel == 0) return false;
  if (h->opt_obs_kernel > 0) return true;
  if (h->block_choice >= 0) return h->block_choice != 0;
  std::vector<uint32_t> xs;
  for (const PauliTerm& t : h->model.terms) xs.push_back(t.x);
  std::sort(xs.begin(), xs.end());
  xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
  const uint32_t gather_block = 256u * obs_amps_per_thread(uint32_t(h->fwd.plan.n_eff));
  double leaving = 0.0;
  std::vector<uint32_t> outs;
  for (uint32_t x : xs) {
    leaving += x >= gather_block ? 1.0 : 0.0;
    outs.push_back(x >> h->opt_obs_block_bits);
  }
  std::sort(outs.begin(), outs.end());
  const double groups = double(std::unique(outs.begin(), outs.end()) - outs.begin());
  const double gather = 2.3 + 0.385 * leaving, block = 0.9 + 0.43 * groups + 0.045 * double(h->model.terms.size());
  h->block_choice = block < gather ? 1 : 0;
  return h->block_choice != 0;
}

// Some term flips two or more qubits, or some group needs a measurement-only pass: measuring in the tiles costs more
// than one sweep of the observable kernel over the final state.
bool wide_observables(const qhbm_engine* h) {
  bool wide = false;
  for (const Pass& p : h->fwd.plan.passes) wide |= p.is_measure_only;
  for (const auto& t : h->model.terms) wide |= __builtin_popcountll(t.x) >= 2;
  return wide;
}

// Several observables (qnn.expectation(states, [op...]), /root/reference/tests/inference/qnn_test.py:187-190): the
// forward sweep runs its lean, paired, measurement-free passes and ONE launch of the block kernel returns every
// <psi|O_t|psi>; a VJP call adds the launch that forms lambda = sum_t upstream_t O_t psi.  Sums of diagonal and
// single-flip terms (TFIM, the Z-string shards of a modular Hamiltonian) keep measuring in the tiles -- the
// Walsh-Hadamard measurement takes hundreds of shards for the price of a few.
constexpr int kMultiValueOps = 64;
bool multi_value_mode(const qhbm_engine* h) {
  // (the values of several observables always come from the block kernel -- the gather kernel has no such mode --,
  // whichever of the two forms lambda)
  if (h->opt_values_from_obs == 0 || h->opt_multi_values == 0 || h->model.n_ops < 2) return false;
  if (gather_multi_forced(h)) return true;
  if (h->fwd.plan.n_eff < kObsBlockBits || h->opt_obs_kernel == 0) return false;
  if (h->model.n_ops > int(kObsMaxValueOps)) return false;
  if (h->opt_multi_values > 0) return true;
  return h->model.n_ops <= kMultiValueOps && h->fwd.plan.passes.size() > 1 && wide_observables(h);
}

// Two to four observables through the gather kernel with a value accumulator per observable
// (apply_observable_kernel<A, OBS_GATHER_MULTI>): ONE launch returns the weighted lambda and every value -- on request only
// ("gather_multi_values" = 1): the per-observable dot products make that launch VALU-bound at 4 x the single-observable
// kernel's time.
bool gather_multi_mode(const qhbm_engine* h) { return gather_multi_forced(h); }

// Forward-only calls: whether the values come from the observable kernel (storing nothing) after lean passes.
bool forward_values_from_observable(const qhbm_engine* h) {
  if (multi_value_mode(h)) return true;
  if (!value_mode(h) || h->opt_fwd_values_obs == 0) return false;
  if (h->opt_fwd_values_obs > 0) return true;
  // measured (scripts/fwd_values_ab.sh): XXZ at 20 qubits - 17 %, 512 random Pauli strings at 24 qubits - 12 %;
  // sums of single-flip and diagonal terms (TFIM) are measured in the tiles at no traffic: + 2 % at 28 and at 16 qubits
  return wide_observables(h) && h->fwd.plan.passes.size() > 1;
}

// `skip_measure`: the caller takes the values from lambda = O psi (value_mode): measurement groups
// are ignored, measurement-only passes and the wide-term kernel are not launched.
int run_forward_chunk(qhbm_engine* h, const int8_t* d_bits, uint32_t s0, uint32_t cs, bool keep_state,
                      hipStream_t stream, bool skip_measure = false) {
  DevicePlan& d = h->fwd;
  const size_t np = d.plan.passes.size();
  bool measure_only_after = !d.plan.global_terms.empty();  // they read the final state from HBM
  for (const Pass& p : d.plan.passes) measure_only_after |= p.is_measure_only;
  for (size_t i = 0; i < np; ++i) {
    const Pass& p = d.plan.passes[i];
    if (skip_measure && p.is_measure_only) continue;
    PassArgs a = d.args[i];
    a.flags = p.flags & (PASS_INIT_BASIS | PASS_GENERAL | PASS_NO_ZERO_FILL);
    if (h->opt_force_general) a.flags |= PASS_GENERAL;
    if (skip_measure) a.flags |= PASS_SKIP_MEASURE;
    if (!p.is_measure_only && (!p.completes_circuit || keep_state || measure_only_after)) a.flags |= PASS_STORE;
    hipEvent_t* ev = timer_begin(h, 0, stream);
    // dense lean passes without a measurement to take: two states per workgroup, tiles in registers
    const bool pair = h->opt_fwd_pair && cs >= 2 && !(a.flags & (PASS_INIT_BASIS | PASS_GENERAL)) && a.zero_mask == 0 &&
                      (skip_measure || p.n_meas_groups == 0) && pass_fwd_pair_supported(p.K);
    if (pair)
      HIPCHK(launch_pass_fwd_pair(p.K, a, cs, h->psi.p, d_bits, h->model.n, s0, d.prog.p, d.tables.p, d.coef.p, stream));
    else
      HIPCHK(launch_pass_fwd(p.K, d.plan.R, a, cs, h->psi.p, d_bits, h->model.n, d.prog.p, d.tables.p, d.coef.p,
                             h->op_scale.p, h->vals64.p, s0, stream));
    timer_end(ev, stream);
  }
  if (!d.plan.global_terms.empty() && !skip_measure)
    HIPCHK(launch_measure_global(h->psi.p, uint32_t(d.plan.n_eff), cs, h->global_terms.p,
                                 uint32_t(d.plan.global_terms.size()), h->op_scale.p, h->vals64.p,
                                 uint32_t(h->model.n_ops), s0, stream));
  return 0;
}

// Expectation values accumulate in fixed point (program.h kValueFracBits) while the passes run ...
int values_begin(qhbm_engine* h, int U, hipStream_t stream) {
  const size_t nv = size_t(U) * size_t(std::max(h->model.n_ops, 1));
  HIPCHK(h->vals64.reserve(nv));
  HIPCHK(launch_zero_fill(h->vals64.p, nv * sizeof(unsigned long long), stream));
  return 0;
}
// ... and are converted to fp32 [U, n_ops] once at the end.
int values_end(qhbm_engine* h, int U, float* d_out, hipStream_t stream) {
  if (!d_out || h->model.n_ops <= 0) return 0;
  HIPCHK(launch_values_from_fixed(h->vals64.p, h->op_inv_scale.p, d_out, uint32_t(U) * uint32_t(h->model.n_ops),
                                  uint32_t(h->model.n_ops), stream));
  return 0;
}

// The state buffers grow with headroom: the number of unique bitstrings of a sampled batch changes
// from step to step, and every new maximum would otherwise free and re-allocate tens of GiB (seconds).
int ensure_state_buffers(qhbm_engine* h, uint32_t cs, bool with_lam) {
  const size_t have = std::min(h->psi.n, with_lam ? h->lam.n : h->psi.n) >> h->fwd.plan.n_eff;
  size_t want = cs;
  if (want > have) {
    const size_t cap = std::max<size_t>(cs, budget_bytes(h) / ((with_lam ? 2 : 1) * state_bytes(h)));
    want = std::min(cap, want + std::max<size_t>(want / 8, 1));
  }
  const size_t amps = want << h->fwd.plan.n_eff;
  if (with_lam && !h->lam.p) {
    // a forward-only call may have grown psi to the whole budget: with lambda beside it the pair must
    // fit the budget again, so psi goes back to its half before lambda is allocated
    const size_t half = std::max<size_t>(amps, (budget_bytes(h) / (2 * state_bytes(h))) << h->fwd.plan.n_eff);
    if (h->psi.n > half) h->psi.release();
  }
  HIPCHK(h->psi.reserve(amps, false));
  if (with_lam) HIPCHK(h->lam.reserve(amps, false));
  return 0;
}

int run_observable_chunk(qhbm_engine* h, uint32_t s0, uint32_t c, const float* d_upstream, bool value_mode,
                         hipStream_t stream, bool store_lambda = true, bool multi_values = false);
int run_values_chunk(qhbm_engine* h, uint32_t row0, uint32_t c, hipStream_t stream);

int forward(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params, float* d_out,
            int shift_gate, double shift, hipStream_t stream) {
  DevicePlan& d = h->fwd;
  h->retained_U = 0;
  HIPCHK(launch_prep_coefs(d.jobs.p, int(d.plan.jobs.size()), d_params, d.coef.p, shift_gate,
                           shift, stream));
  HIPCHK(launch_combine_diag(d.coef.p, d.rec_offsets.p, int(d.plan.record_offsets.size()), 1u, 0u, stream));
  if (int rc = values_begin(h, U, stream)) return rc;
  const uint32_t cs = chunk_states(h, U);
  if (int rc = ensure_state_buffers(h, cs, false)) return rc;
  // One observable and a state that passes through HBM anyway: the lean (paired, measurement-free) passes and one
  // sweep of the lambda = O psi kernel for <psi|O|psi> -- nothing stored -- beat the measuring passes (config 3:
  // 148 -> 124 ms per 4096 states).  A single-pass plan measures in its tile and never writes the state.
  const bool from_obs = forward_values_from_observable(h);
  for (uint32_t s0 = 0; s0 < uint32_t(U); s0 += cs) {
    const uint32_t c = std::min<uint32_t>(cs, uint32_t(U) - s0);
    if (int rc = run_forward_chunk(h, d_bits, s0, c, from_obs, stream, from_obs)) return rc;
    if (from_obs)
      if (int rc = run_values_chunk(h, s0, c, stream)) return rc;
  }
  return values_end(h, U, d_out, stream);
}

// One state per XCD at a time (1) or every XCD an eighth of each state (0)?  Measured on the block kernel, config 4
// (24 qubits): 61.1 against 53.6 ms per 32 states -- a 128-MiB state that all eight XCDs read sits in the Infinity
// Cache once; the gather kernel at config 3 (20 qubits): one state per XCD wins (3.7 against 5.1 fabric reads of the
// state).  By size: states of 64 MiB and more are shared by the XCDs.
bool observable_xcd_states(const qhbm_engine* h) {
  if (h->opt_obs_xcd_states >= 0) return h->opt_obs_xcd_states != 0;
  return h->fwd.plan.n_eff < 23;
}

// lambda = O psi for the chunk in the workspace.  value_mode (a single observable): unweighted, and
// <psi|O|psi> goes to the fixed-point value accumulators -- the forward sweep measured nothing.
// multi_values (gather_multi_mode): the gather kernel with a value accumulator per observable -- the weighted lambda (if
// stored) and every <psi|O_t|psi> from ONE launch.
int run_observable_chunk(qhbm_engine* h, uint32_t s0, uint32_t c, const float* d_upstream, bool value_mode,
                         hipStream_t stream, bool store_lambda, bool multi_values) {
  const uint32_t n_eff = uint32_t(h->fwd.plan.n_eff);
  if (multi_values) {
    HIPCHK(h->value_part.reserve(observable_value_parts(n_eff, c) * size_t(h->model.n_ops)));
    hipEvent_t* ev = timer_begin(h, 2, stream);
    HIPCHK(launch_apply_observable(h->psi.p, store_lambda ? h->lam.p : nullptr, n_eff, c, h->terms.p,
                                   h->n_gather_terms, h->obs_groups.p, h->n_obs_groups,
                                   store_lambda ? d_upstream : nullptr, uint32_t(h->model.n_ops), s0, h->op_scale.p,
                                   h->vals64.p, h->value_part.p, observable_xcd_states(h), stream, true));
    timer_end(ev, stream);
    return 0;
  }
  if (value_mode)
    HIPCHK(h->value_part.reserve(block_kernel(h) ? observable_blocks_value_parts(n_eff, c, 1u, h->opt_obs_block_bits) : observable_value_parts(n_eff, c)));
  hipEvent_t* ev = timer_begin(h, 2, stream);
  if (block_kernel(h)) {
    const int mode = !value_mode ? OBS_LAMBDA : (store_lambda ? OBS_LAMBDA_VALUE : OBS_VALUES);
    HIPCHK(launch_observable_blocks(mode, obs_block_shape(h), h->psi.p, store_lambda ? h->lam.p : nullptr, n_eff, c, h->obs_bterms.p,
                                    h->obs_bgroups.p, h->n_obs_bgroups, d_upstream, uint32_t(h->model.n_ops), s0,
                                    h->op_scale.p, value_mode ? h->vals64.p : nullptr, h->value_part.p,
                                    observable_xcd_states(h), stream));
  } else {
    ObsFarLaunch far[3];
    for (int f = 0; f < h->n_far; ++f)
      far[f] = ObsFarLaunch{h->far[f].terms.p, h->far[f].n_terms, h->far[f].groups.p, h->far[f].n_groups, h->far[f].far_hi};
    HIPCHK(launch_apply_observable(h->psi.p, store_lambda ? h->lam.p : nullptr, n_eff, c, h->terms.p,
                                   h->n_gather_terms, h->obs_groups.p, h->n_obs_groups, d_upstream,
                                   uint32_t(h->model.n_ops), s0, h->op_scale.p, value_mode ? h->vals64.p : nullptr,
                                   h->value_part.p, observable_xcd_states(h), stream, false, far, h->n_far));
  }
  timer_end(ev, stream);
  return 0;
}

// <psi|O_t|psi> of every observable for `c` final states in psi (rows row0 .. row0 + c of the fixed-point accumulators),
// nothing stored: one observable through either kernel, several through the block kernel (multi_value_mode).
int run_values_chunk(qhbm_engine* h, uint32_t row0, uint32_t c, hipStream_t stream) {
  if (h->model.n_ops == 1) return run_observable_chunk(h, row0, c, nullptr, true, stream, false);
  if (gather_multi_mode(h)) return run_observable_chunk(h, row0, c, nullptr, false, stream, false, true);
  const uint32_t n_eff = uint32_t(h->fwd.plan.n_eff);
  HIPCHK(h->value_part.reserve(observable_blocks_value_parts(n_eff, c, uint32_t(h->model.n_ops), h->opt_obs_block_bits)));
  hipEvent_t* ev = timer_begin(h, 2, stream);
  HIPCHK(launch_observable_blocks(OBS_VALUES_MULTI, obs_block_shape(h), h->psi.p, nullptr, n_eff, c, h->obs_bterms.p, h->obs_bgroups.p,
                                  h->n_obs_bgroups, nullptr, uint32_t(h->model.n_ops), row0, h->op_scale.p, h->vals64.p,
                                  h->value_part.p, observable_xcd_states(h), stream));
  timer_end(ev, stream);
  return 0;
}

// The backward passes of a chunk whose (psi, lambda) pair is in the workspace.
int run_adjoint_chunk(qhbm_engine* h, const int8_t* d_bits, uint32_t s0, uint32_t c, hipStream_t stream) {
  DevicePlan& b = h->adj;
  const uint32_t n_slots = uint32_t(b.plan.slot_gate.size());
  size_t rows = 0;  // tile_grad: one row of the pass's slots per workgroup
  for (const PassArgs& ba : b.args)
    rows = std::max(rows, ((size_t(c) << ba.n_free) + reduce_tiles_scratch_rows(c, size_t(1) << ba.n_free)) *
                              std::max<uint32_t>(ba.n_slots, 1));
  HIPCHK(h->tile_grad.reserve(rows));
  for (size_t i = 0; i < b.plan.passes.size(); ++i) {
    hipEvent_t* ev = timer_begin(h, 1, stream);
    PassArgs ba = b.args[i];
    if (h->opt_force_general) ba.flags |= PASS_GENERAL;
    HIPCHK(launch_pass_adj(b.plan.K, h->opt_adj_exchange != 0, ba, c, h->psi.p, h->lam.p, d_bits, h->model.n, b.prog.p,
                           b.tables.p, b.coef.p, h->tile_grad.p, s0, stream));
    timer_end(ev, stream);
    // tiles of a state are added in tile order (bit-reproducible, no atomics)
    HIPCHK(launch_reduce_tiles(h->tile_grad.p, c, 1u << ba.n_free, ba.n_slots, h->state_grad.p, n_slots, ba.slot_base,
                               s0, stream));
  }
  return 0;
}

uint32_t adjoint_chunk_states(qhbm_engine* h, int U) {
  uint32_t cs = chunk_states(h, U);
  if (h->opt_chunk <= 0)  // two buffers per state
    cs = uint32_t(std::min<size_t>(std::min<size_t>(size_t(U), 65535),
                                   std::max<size_t>(1, budget_bytes(h) / (2 * state_bytes(h)))));
  return cs;
}

// values + per-state gradient slots (adjoint) for a given upstream.
int adjoint_sweep(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params,
                  const float* d_upstream, float* d_out_vals, hipStream_t stream) {
  DevicePlan& f = h->fwd;
  DevicePlan& b = h->adj;
  h->retained_U = 0;
  h->state_grad_U = 0;
  const uint32_t n_slots = uint32_t(b.plan.slot_gate.size());
  HIPCHK(launch_prep_coefs(f.jobs.p, int(f.plan.jobs.size()), d_params, f.coef.p, -1, 0.0, stream));
  HIPCHK(launch_combine_diag(f.coef.p, f.rec_offsets.p, int(f.plan.record_offsets.size()), 1u, 0u, stream));
  HIPCHK(launch_prep_coefs(b.jobs.p, int(b.plan.jobs.size()), d_params, b.coef.p, -1, 0.0, stream));
  HIPCHK(launch_combine_diag(b.coef.p, b.rec_offsets.p, int(b.plan.record_offsets.size()), 1u, 0u, stream));
  if (int rc = values_begin(h, U, stream)) return rc;
  HIPCHK(h->state_grad.reserve(size_t(U) * std::max<uint32_t>(n_slots, 1)));
  HIPCHK(launch_zero_fill(h->state_grad.p, size_t(U) * std::max<uint32_t>(n_slots, 1) * sizeof(float), stream));
  const uint32_t cs = adjoint_chunk_states(h, U);
  if (int rc = ensure_state_buffers(h, cs, true)) return rc;
  const bool vm = value_mode(h), mv = multi_value_mode(h);
  for (uint32_t s0 = 0; s0 < uint32_t(U); s0 += cs) {
    const uint32_t c = std::min<uint32_t>(cs, uint32_t(U) - s0);
    if (int rc = run_forward_chunk(h, d_bits, s0, c, true, stream, vm || mv)) return rc;
    const bool gm = mv && gather_multi_mode(h);  // 2..4 observables on the gather kernel: values AND lambda from one launch
    if (mv && !gm)  // several observables: their values from one launch over the final states, lambda (weighted) from the next
      if (int rc = run_values_chunk(h, s0, c, stream)) return rc;
    if (int rc = run_observable_chunk(h, s0, c, d_upstream, vm, stream, true, gm)) return rc;
    if (int rc = run_adjoint_chunk(h, d_bits, s0, c, stream)) return rc;
  }
  // value mode ran the sweep on the unweighted lambda: the upstream weight goes onto the gradient rows
  if (vm) HIPCHK(launch_scale_rows(h->state_grad.p, uint32_t(U), std::max<uint32_t>(n_slots, 1), d_upstream, stream));
  return values_end(h, U, d_out_vals, stream);
}

int check_call(qhbm_engine* h, int U) {
  if (!h) return 1;
  if (int rc = need_device(h)) return rc;
  if (U < 0) return fail(h, "negative batch size");
  if (h->model.n_ops <= 0) return fail(h, "qhbm_set_observables has not been called");
  return upload_model(h);
}

// Parameter shift with a shared prefix.  A shifted program differs from the base program in the coefficients of ONE gate;
// the passes in front of the first pass that reads them compute, state by state, the bits the base program computes.
// first_dependent_pass()[g] = that pass for gate g (the pass count for a gate no record depends on); `first_measuring` =
// the first pass with a measurement op (a program must not start behind it: a pass measures into the program's own
// accumulators).  Record ranges come from the pass programs (OP_ROUND: n_instances records from its first record;
// OP_GATE2: a 4 x 4 matrix), jobs from the plan (CoefJob::gate writes at CoefJob::out_off).
std::vector<int> first_dependent_pass(const Plan& plan, size_t n_gates, int* first_measuring) {
  const RecordLayout L(plan.R, false);
  struct Range { uint32_t lo, hi; int pass; };
  std::vector<Range> ranges;
  const int n_pass = int(plan.passes.size());
  *first_measuring = n_pass;
  for (int i = 0; i < n_pass; ++i) {
    const std::vector<uint32_t>& prog = plan.passes[size_t(i)].prog;
    size_t pc = 0;
    while (pc < prog.size()) {
      const uint32_t w0 = prog[pc], opc = w0 & 0xffu;
      if (opc == OP_END) break;
      if (opc == OP_ROUND) {
        const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8, rec = prog[pc + 2];
        ranges.push_back(Range{rec, rec + n_inst * uint32_t(L.words()), i});
        pc += kRoundWords;
      } else if (opc == OP_GATE2) {
        ranges.push_back(Range{prog[pc + 2], prog[pc + 2] + 32u, i});
        pc += kGate2Words;
      } else if (opc == OP_MEASURE_WHT) {
        *first_measuring = std::min(*first_measuring, i);
        pc += size_t(kWhtHeaderWords) + size_t(w0 >> 8) * kMeasTermWords;
      } else if (opc == OP_MEASURE) {
        const uint32_t n_groups = w0 >> 8;
        if (n_groups) *first_measuring = std::min(*first_measuring, i);
        pc += 1;
        for (uint32_t g = 0; g < n_groups && pc + 1 < prog.size(); ++g) pc += 2u + size_t(prog[pc + 1]) * kMeasTermWords;
      } else {
        return std::vector<int>(n_gates, 0);  // an opcode this walk does not know: share nothing
      }
    }
  }
  std::vector<int> first(n_gates, n_pass);
  for (const CoefJob& j : plan.jobs) {
    if (j.gate < 0 || size_t(j.gate) >= n_gates) continue;
    int pass = 0;  // (a job whose record no pass claims: share nothing for its gate)
    for (const Range& r : ranges)
      if (uint32_t(j.out_off) >= r.lo && uint32_t(j.out_off) < r.hi) { pass = r.pass; break; }
    first[size_t(j.gate)] = std::min(first[size_t(j.gate)], pass);
  }
  return first;
}

}  // namespace

// ================================================================================
extern "C" {

int qhbm_abi_version(void) { return QHBM_ABI_VERSION; }

int qhbm_create(int device, qhbm_engine** out) {
  if (!out) return fail(nullptr, "out is NULL");
  std::unique_ptr<qhbm_engine> h(new qhbm_engine());
  h->device = device;
  if (const char* bb = std::getenv("QHBM_OBS_BLOCK_BITS"))  // (A/B runs of whole programs: the options' defaults)
    if (std::atoi(bb) == kObsBlockBits || std::atoi(bb) == kObsBlockBitsSmall) h->opt_obs_block_bits = std::atoi(bb);
  if (const char* sr = std::getenv("QHBM_OBS_SPLIT_ROWS")) h->opt_obs_split_rows = std::atoi(sr) != 0;
  if (device >= 0) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device >= count)
      return fail(nullptr, std::string("no HIP device ") + std::to_string(device) + ": " +
                               (e != hipSuccess ? hipGetErrorString(e) : "index out of range"));
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, std::string("hipSetDevice: ") + hipGetErrorString(e));
  }
  *out = h.release();
  return 0;
}

void qhbm_destroy(qhbm_engine* h) {
  if (!h) return;
  if (h->device >= 0) (void)hipSetDevice(h->device);
  for (auto& ev : h->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  for (auto& ev : h->free_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  delete h;
}

const char* qhbm_last_error(const qhbm_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int qhbm_set_circuit(qhbm_engine* h, int n_qubits, int n_gates, const qhbm_gate* gates, int n_params) {
  if (!h) return 1;
  if (n_qubits < 1 || n_qubits > 31) return fail(h, "n_qubits must be in [1, 31]");
  if (n_gates < 0 || n_params < 0 || (n_gates > 0 && !gates)) return fail(h, "bad gate list");
  static_assert(sizeof(qhbm_gate) == sizeof(Gate), "ABI gate layout");
  Model m = h->model;
  if (m.n != n_qubits) { m.terms.clear(); m.n_ops = 0; }
  m.n = n_qubits;
  m.n_params = n_params;
  m.param_frozen.clear();  // (a mask belongs to the circuit it was set for)
  m.gates.resize(size_t(n_gates));
  if (n_gates) std::memcpy(m.gates.data(), gates, size_t(n_gates) * sizeof(Gate));
  for (const Gate& G : m.gates)
    if (!std::isfinite(G.global_shift)) return fail(h, "gate with a non-finite global_shift");
  Plan probe;
  std::string err;
  Model no_obs = m;
  no_obs.terms.clear();
  no_obs.n_ops = 0;
  if (!build_plan(no_obs, h->opt_tile, h->opt_round, false, &probe, &err)) return fail(h, err);
  h->model = std::move(m);
  h->have_circuit = true;
  h->plans_valid = false;
  h->block_choice = -1;
  return 0;
}

int qhbm_set_gradient_mask(qhbm_engine* h, const uint8_t* needs_grad, int n_params) {
  if (!h) return 1;
  if (!h->have_circuit) return fail(h, "call qhbm_set_circuit first");
  std::vector<char> frozen;
  if (needs_grad) {
    if (n_params != h->model.n_params) return fail(h, "gradient mask: n_params differs from the circuit's");
    frozen.resize(size_t(n_params));
    bool any = false;
    for (int i = 0; i < n_params; ++i) any |= (frozen[size_t(i)] = needs_grad[i] ? 0 : 1) != 0;
    if (!any) frozen.clear();
  }
  if (frozen != h->model.param_frozen) {
    // the backward plan and the shift tables depend on the mask; the forward plan does not
    if (h->plans_valid && h->adj_valid) {  // keep the plan of the mask that is leaving (at most four)
      if (h->adj_cache.size() >= 4) h->adj_cache.erase(h->adj_cache.begin());
      h->adj_cache.emplace_back(h->model.param_frozen, std::unique_ptr<DevicePlan>(new DevicePlan()));
      h->adj_cache.back().second->swap(h->adj);
    }
    h->model.param_frozen = std::move(frozen);
    h->adj_valid = false;
    for (size_t i = 0; i < h->adj_cache.size(); ++i)
      if (h->plans_valid && h->adj_cache[i].first == h->model.param_frozen) {  // seen before: swap it back in
        h->adj.swap(*h->adj_cache[i].second);
        h->adj_cache.erase(h->adj_cache.begin() + long(i));
        h->adj_valid = true;
        h->model_uploaded = false;  // (the parameter -> slot tables follow the plan: a few KiB, no planning)
        break;
      }
    h->shift_ready = false;
    h->retained_U = 0;
    h->state_grad_U = 0;
  }
  return 0;
}

int qhbm_set_observables(qhbm_engine* h, int n_ops, const int32_t* term_offsets, const float* coeffs,
                         const uint64_t* x_masks, const uint64_t* z_masks) {
  if (!h) return 1;
  if (!h->have_circuit) return fail(h, "call qhbm_set_circuit first");
  if (n_ops < 1 || n_ops > kMaxOps || !term_offsets) return fail(h, "n_ops must be in [1, 1024]");
  const int n = h->model.n;
  std::vector<PauliTerm> terms;
  for (int k = 0; k < n_ops; ++k) {
    if (term_offsets[k + 1] < term_offsets[k]) return fail(h, "term_offsets must be non-decreasing");
    for (int j = term_offsets[k]; j < term_offsets[k + 1]; ++j) {
      const uint64_t x = x_masks[j], z = z_masks[j];
      if ((x | z) >> n) return fail(h, "Pauli mask addresses a qubit >= n_qubits");
      PauliTerm t;
      t.coeff = coeffs[j];
      t.x = t.z = 0;
      for (int q = 0; q < n; ++q) {
        if (x >> q & 1) t.x |= 1u << (n - 1 - q);
        if (z >> q & 1) t.z |= 1u << (n - 1 - q);
      }
      t.ny = __builtin_popcountll(x & z);
      t.op = k;
      terms.push_back(t);
    }
  }
  // fixed-point scale of every op's accumulator: partial sums are bounded by B = sum |c_k|
  h->h_op_scale.assign(size_t(n_ops), 0.f);
  h->h_op_inv_scale.assign(size_t(n_ops), 0.f);
  for (int k = 0; k < n_ops; ++k) {
    double bound = 0.0;
    for (int j = term_offsets[k]; j < term_offsets[k + 1]; ++j) bound += std::fabs(double(coeffs[j]));
    int e = 0;
    if (bound > 0.0) (void)std::frexp(bound, &e);  // bound <= 2^e
    e = std::max(-60, std::min(60, e));
    h->h_op_scale[size_t(k)] = float(std::ldexp(1.0, kValueFracBits - e));
    h->h_op_inv_scale[size_t(k)] = float(std::ldexp(1.0, e - kValueFracBits));
  }
  h->model.n_ops = n_ops;
  h->model.terms = std::move(terms);
  h->plans_valid = false;
  h->block_choice = -1;
  h->terms.release();
  return 0;
}

int qhbm_set_option(qhbm_engine* h, const char* name, int64_t value) {
  if (!h || !name) return 1;
  const std::string k(name);
  if (k == "tile_qubits") { h->opt_tile = int(value); h->plans_valid = false; h->block_choice = -1; }
  else if (k == "round_qubits") { h->opt_round = int(value); h->plans_valid = false; }
  else if (k == "force_general_kernels") h->opt_force_general = int(value);
  else if (k == "full_diag_threshold") { h->opt_full_fwd = int(value); h->plans_valid = false; }
  else if (k == "adjoint_full_diag_threshold") { h->opt_full_adj = int(value); h->plans_valid = false; }
  else if (k == "adjoint_tile_qubits") { h->opt_adj_tile = int(value); h->plans_valid = false; }
  else if (k == "adjoint_exchange") { h->opt_adj_exchange = int(value); h->plans_valid = false; }
  else if (k == "adjoint_relabel") { h->opt_adj_relabel = int(value); h->plans_valid = false; }
  else if (k == "forward_pairs") h->opt_fwd_pair = int(value);
  else if (k == "shift_prefix_sharing") { h->opt_shift_prefix = int(value); h->shift_ready = false; }
  else if (k == "forward_values_from_observable") h->opt_fwd_values_obs = int(value);
  else if (k == "adjoint_stop_early") { h->opt_adj_stop_early = int(value); h->plans_valid = false; }
  else if (k == "adjoint_plan_search") { h->opt_adj_plan_search = int(value); h->plans_valid = false; }
  else if (k == "wide_last_pass") { h->opt_wide_last = int(value); h->plans_valid = false; }
  else if (k == "observable_xcd_states") h->opt_obs_xcd_states = int(value);
  else if (k == "observable_kernel") { h->opt_obs_kernel = int(value); h->block_choice = -1; }
  else if (k == "observable_split_rows") h->opt_obs_split_rows = value != 0;
  else if (k == "observable_block_bits") {
    if (value != kObsBlockBits && value != kObsBlockBitsSmall) return fail(h, "observable_block_bits: 12 or 13");
    h->opt_obs_block_bits = int(value); h->block_choice = -1; h->terms.release(); h->model_uploaded = false;
  }
  else if (k == "multi_observable_values") h->opt_multi_values = int(value);
  else if (k == "gather_multi_values") h->opt_gather_multi = int(value);
  else if (k == "observable_far_windows") { h->opt_far_windows = int(value); h->terms.release(); h->model_uploaded = false; }
  else if (k == "measure_tile_qubits") { h->opt_meas_tile = int(value); h->plans_valid = false; }
  else if (k == "values_from_observable") h->opt_values_from_obs = int(value);
  else if (k == "cph_wave_bits") { h->opt_cph_wave_bits = int(value); h->plans_valid = false; }
  else if (k == "chunk_states") h->opt_chunk = value;
  else if (k == "workspace_budget_mb") h->opt_budget_mb = std::max<int64_t>(0, value);  // 0 = default
  else if (k == "profile_events") h->opt_profile = int(value);
  else return fail(h, "unknown option '" + k + "'");
  return 0;
}

int qhbm_workspace_bytes(qhbm_engine* h, int U, int with_vjp, size_t* out) {
  if (!h || !out) return 1;
  if (int rc = build_plans(h)) return rc;
  const uint32_t cs = with_vjp ? adjoint_chunk_states(h, U) : chunk_states(h, U);  // as the calls chunk
  size_t b = size_t(cs) * state_bytes(h) * (with_vjp ? 2 : 1);
  if (with_vjp) {
    b += size_t(U) * h->adj.plan.slot_gate.size() * sizeof(float);
    size_t rows = 0;  // per-tile gradient rows of the widest adjoint pass
    for (const Pass& p : h->adj.plan.passes)
      rows = std::max(rows, ((size_t(cs) << p.nonlocal_pos.size()) +
                             reduce_tiles_scratch_rows(cs, size_t(1) << p.nonlocal_pos.size())) * size_t(std::max(p.n_slots, 1)));
    b += rows * sizeof(float);
  }
  b += size_t(U) * size_t(std::max(h->model.n_ops, 1)) * sizeof(unsigned long long);
  *out = b;
  return 0;
}

int qhbm_allocated_bytes(qhbm_engine* h, size_t* out) {
  if (!h || !out) return 1;
  *out = own_bytes(h);
  return 0;
}

int qhbm_retained_states(qhbm_engine* h, int* out_U) {
  if (!h || !out_U) return 1;
  *out_U = h->retained_U;
  return 0;
}

int qhbm_expectation(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params, float* d_out,
                     void* stream) {
  if (int rc = check_call(h, U)) return rc;
  if (U == 0) return 0;
  return forward(h, d_bits, U, d_params, d_out, -1, 0.0, static_cast<hipStream_t>(stream));
}

int qhbm_expectation_retain(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params, float* d_out,
                            void* stream) {
  if (int rc = check_call(h, U)) return rc;
  if (U == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (adjoint_chunk_states(h, U) < uint32_t(U))  // the batch does not fit one backward chunk: nothing kept
    return forward(h, d_bits, U, d_params, d_out, -1, 0.0, s);
  DevicePlan& d = h->fwd;
  h->retained_U = 0;
  HIPCHK(launch_prep_coefs(d.jobs.p, int(d.plan.jobs.size()), d_params, d.coef.p, -1, 0.0, s));
  HIPCHK(launch_combine_diag(d.coef.p, d.rec_offsets.p, int(d.plan.record_offsets.size()), 1u, 0u, s));
  if (int rc = values_begin(h, U, s)) return rc;
  if (int rc = ensure_state_buffers(h, uint32_t(U), true)) return rc;  // psi AND lambda, so psi is not moved later
  const bool vm = value_mode(h), mv = multi_value_mode(h);
  if (int rc = run_forward_chunk(h, d_bits, 0, uint32_t(U), true, s, vm || mv)) return rc;
  if (vm) {
    if (int rc = run_observable_chunk(h, 0, uint32_t(U), nullptr, true, s)) return rc;
  } else if (mv) {
    if (int rc = run_values_chunk(h, 0, uint32_t(U), s)) return rc;
  }
  if (int rc = values_end(h, U, d_out, s)) return rc;
  h->retained_U = U;
  h->retained_mu = vm;
  return 0;
}

int qhbm_expectation_vjp_retained(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params,
                                  const float* d_upstream, float* d_grad, void* stream) {
  if (int rc = check_call(h, U)) return rc;
  if (U <= 0 || h->retained_U != U) return fail(h, "no retained forward state for this batch");
  hipStream_t s = static_cast<hipStream_t>(stream);
  DevicePlan& b = h->adj;
  h->retained_U = 0;  // the backward sweep un-applies psi in place: the state is consumed
  h->state_grad_U = 0;
  const uint32_t n_slots = uint32_t(b.plan.slot_gate.size());
  HIPCHK(launch_prep_coefs(b.jobs.p, int(b.plan.jobs.size()), d_params, b.coef.p, -1, 0.0, s));
  HIPCHK(launch_combine_diag(b.coef.p, b.rec_offsets.p, int(b.plan.record_offsets.size()), 1u, 0u, s));
  HIPCHK(h->state_grad.reserve(size_t(U) * std::max<uint32_t>(n_slots, 1)));
  HIPCHK(launch_zero_fill(h->state_grad.p, size_t(U) * std::max<uint32_t>(n_slots, 1) * sizeof(float), s));
  if (h->retained_mu) {  // lambda = O psi (unweighted) was computed with the values: weight the rows instead
    if (int rc = run_adjoint_chunk(h, d_bits, 0, uint32_t(U), s)) return rc;
    HIPCHK(launch_scale_rows(h->state_grad.p, uint32_t(U), std::max<uint32_t>(n_slots, 1), d_upstream, s));
  } else {
    if (int rc = run_observable_chunk(h, 0, uint32_t(U), d_upstream, false, s)) return rc;
    if (int rc = run_adjoint_chunk(h, d_bits, 0, uint32_t(U), s)) return rc;
  }
  HIPCHK(launch_reduce_grad(h->state_grad.p, uint32_t(U), n_slots, h->param_slot_begin.p, h->param_slots.p,
                            h->slot_factor.p, d_grad, h->model.n_params, 0, s));
  h->state_grad_U = U;
  return 0;
}

int qhbm_state_gradients(qhbm_engine* h, int U, float* d_rows, void* stream) {
  if (!h || !d_rows) return 1;
  if (int rc = need_device(h)) return rc;
  if (U <= 0 || U != h->state_grad_U || !h->plans_valid || !h->adj_valid)
    return fail(h, "qhbm_state_gradients: the last call was not an adjoint VJP on this many states");
  HIPCHK(launch_scatter_jac(h->state_grad.p, uint32_t(U), uint32_t(h->adj.plan.slot_gate.size()),
                            h->param_slot_begin.p, h->param_slots.p, h->slot_factor.p, d_rows, 1u, 0u,
                            uint32_t(h->model.n_params), static_cast<hipStream_t>(stream)));
  return 0;
}

int qhbm_statevector(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params,
                     void* d_out_states, void* stream) {
  if (!h) return 1;
  if (int rc = need_device(h)) return rc;
  if (U < 0) return fail(h, "negative batch size");
  if (int rc = upload_model(h)) return rc;
  if (U == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DevicePlan& d = h->fwd;
  h->retained_U = 0;
  HIPCHK(launch_prep_coefs(d.jobs.p, int(d.plan.jobs.size()), d_params, d.coef.p, -1, 0.0, s));
  HIPCHK(launch_combine_diag(d.coef.p, d.rec_offsets.p, int(d.plan.record_offsets.size()), 1u, 0u, s));
  // expectation values of installed observables are a by-product; they stay in the fixed-point scratch
  if (int rc = values_begin(h, U, s)) return rc;
  const uint32_t cs = chunk_states(h, U);
  if (int rc = ensure_state_buffers(h, cs, false)) return rc;
  const size_t row = size_t(8) << h->model.n, pitch = size_t(8) << d.plan.n_eff;
  for (uint32_t s0 = 0; s0 < uint32_t(U); s0 += cs) {
    const uint32_t c = std::min<uint32_t>(cs, uint32_t(U) - s0);
    if (int rc = run_forward_chunk(h, d_bits, s0, c, true, s)) return rc;
    // idle padding qubits (n < 10) are the high index bits and stay |0>: keep the first 2^n amplitudes
    HIPCHK(hipMemcpy2DAsync(static_cast<char*>(d_out_states) + size_t(s0) * row, row, h->psi.p, pitch, row, c,
                            hipMemcpyDeviceToDevice, s));
  }
  // restore the global phase the kernels leave out: cirq's e^{i pi t / 2} per X**t / Y**t and every
  // gate's exp(i pi t global_shift)
  HIPCHK(h->phase_cs.reserve(2));
  HIPCHK(launch_global_phase(d.jobs.p, int(d.plan.jobs.size()), h->shift_phases.p, h->n_shift_phases, d_params,
                             h->phase_cs.p, s));
  HIPCHK(launch_scale_states(static_cast<float2*>(d_out_states), size_t(U) << h->model.n, h->phase_cs.p, s));
  return 0;
}

int qhbm_parity_energy(const int8_t* d_bits, int64_t n_rows, int n_bits, const uint64_t* d_masks,
                       const float* d_thetas, int n_terms, float* d_energy, void* stream) {
  if (n_rows < 0 || n_terms < 0) return fail(nullptr, "negative size");
  if (n_bits < 1 || n_bits > 64) return fail(nullptr, "n_bits must be in [1, 64]");
  hipError_t e = launch_parity_energy(d_bits, n_rows, n_bits, d_masks, d_thetas, n_terms, d_energy,
                                      static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(nullptr, std::string("qhbm_parity_energy: ") + hipGetErrorString(e));
  return 0;
}

int qhbm_parity_energy_vjp(const int8_t* d_bits, int64_t n_rows, int n_bits, const uint64_t* d_masks,
                           int n_terms, const float* d_weights, float* d_grad, void* stream) {
  if (n_rows < 0 || n_terms < 0) return fail(nullptr, "negative size");
  if (n_bits < 1 || n_bits > 64) return fail(nullptr, "n_bits must be in [1, 64]");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = n_terms ? launch_zero_fill(d_grad, size_t(n_terms) * sizeof(float), s) : hipSuccess;
  if (e == hipSuccess) e = launch_parity_energy_vjp(d_bits, n_rows, n_bits, d_masks, n_terms, d_weights, d_grad, s);
  if (e != hipSuccess) return fail(nullptr, std::string("qhbm_parity_energy_vjp: ") + hipGetErrorString(e));
  return 0;
}

int qhbm_sample(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params, int n_shots,
                uint64_t seed, int shift_gate, double shift, int8_t* d_out_samples, void* stream) {
  if (!h) return 1;
  if (int rc = need_device(h)) return rc;
  if (U < 0 || n_shots < 0) return fail(h, "negative batch size or shot count");
  if (shift_gate >= int(h->model.gates.size())) return fail(h, "shift_gate out of range");
  if (shift_gate < 0) shift_gate = -1;  // any negative value = the unshifted circuit (never the sentinel of a lowered fixed op)
  else if (shift != 0.0 && h->model.gates[size_t(shift_gate)].param_idx < 0)
    return fail(h, "shift_gate addresses a gate with a constant exponent: only parametrised gates have shifted programs "
                   "(tfq ParameterShift.get_gradient_circuits shifts symbols), and a lowered constant gate cannot be shifted");
  if (int rc = upload_model(h)) return rc;
  if (U == 0 || n_shots == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DevicePlan& d = h->fwd;
  h->retained_U = 0;
  HIPCHK(launch_prep_coefs(d.jobs.p, int(d.plan.jobs.size()), d_params, d.coef.p, shift_gate, shift, s));
  HIPCHK(launch_combine_diag(d.coef.p, d.rec_offsets.p, int(d.plan.record_offsets.size()), 1u, 0u, s));
  if (int rc = values_begin(h, U, s)) return rc;
  const uint32_t cs = std::min<uint32_t>(chunk_states(h, U), 65535u);
  if (int rc = ensure_state_buffers(h, cs, false)) return rc;
  const uint32_t n_eff = uint32_t(d.plan.n_eff);
  HIPCHK(h->block_cum.reserve(size_t(cs) << (n_eff - 10)));
  for (uint32_t s0 = 0; s0 < uint32_t(U); s0 += cs) {
    const uint32_t c = std::min<uint32_t>(cs, uint32_t(U) - s0);
    if (int rc = run_forward_chunk(h, d_bits, s0, c, true, s)) return rc;
    // shots are drawn in slices of <= 65535 (grid.x); the shot index feeds the counter RNG
    HIPCHK(launch_sample(h->psi.p, n_eff, h->model.n, c, h->block_cum.p, uint32_t(n_shots), seed, s0,
                         d_out_samples, s));
  }
  return 0;
}

int qhbm_sample_counts(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params, int n_programs,
                       const int32_t* shift_gates, const float* shifts, int n_shots, uint64_t seed,
                       int32_t* d_out_counts, void* stream) {
  if (!h) return 1;
  if (int rc = need_device(h)) return rc;
  if (U < 0 || n_shots < 0 || n_programs < 0) return fail(h, "negative batch size, program or shot count");
  if (n_programs && (!shift_gates || !shifts)) return fail(h, "shift_gates / shifts are NULL");
  if (h->model.n > 24) return fail(h, "qhbm_sample_counts keeps 2^n counters per (program, state): n_qubits <= 24; use qhbm_sample");
  for (int q = 0; q < n_programs; ++q)
    if (shift_gates[q] >= int(h->model.gates.size())) return fail(h, "shift_gates entry out of range");
  for (int q = 0; q < n_programs; ++q)
    if (shift_gates[q] >= 0 && shifts[q] != 0.f && h->model.gates[size_t(shift_gates[q])].param_idx < 0)
      return fail(h, "shift_gates entry addresses a gate with a constant exponent: only parametrised gates have shifted programs");
  if (int rc = upload_model(h)) return rc;
  if (U == 0 || n_programs == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DevicePlan& d = h->fwd;
  h->retained_U = 0;
  h->state_grad_U = 0;
  h->shift_ready = false;  // the shift tables below replace those of the parameter-shift VJP
  {
    std::vector<int> sg(shift_gates, shift_gates + n_programs);
    for (int& g : sg) g = std::max(g, -1);  // any negative value = the unshifted circuit
    std::vector<float> sv(shifts, shifts + n_programs);
    HIPCHK(hipStreamSynchronize(s));  // synchronous copies into buffers an earlier call on this stream may still read
    HIPCHK(h->shift_gates.upload(sg));
    HIPCHK(h->shift_vals.upload(sv));
  }
  const uint32_t n_prog = uint32_t(n_programs), n_eff = uint32_t(d.plan.n_eff);
  const uint32_t stride = uint32_t((d.plan.coef_init.size() + 64 + 63) / 64 * 64);
  // launch-set geometry as in the parameter-shift VJP: Uc states x Pc programs per set
  size_t cap = std::max<size_t>(1, budget_bytes(h) / state_bytes(h));
  uint32_t max_nl = 0;
  for (const PassArgs& a : d.args) max_nl = std::max(max_nl, a.n_nonlocal);
  cap = std::min<size_t>(cap, (size_t(1) << 30) >> max_nl);
  cap = std::min<size_t>(cap, (size_t(2) << 30) / (size_t(stride) * sizeof(float)));
  cap = std::min<size_t>(cap, 65535);  // grid.y of the per-element kernels
  if (h->opt_chunk > 0) cap = std::min<size_t>(cap, size_t(h->opt_chunk));
  const uint32_t Uc = uint32_t(std::min<size_t>(size_t(U), cap));
  const uint32_t Pc = uint32_t(std::max<size_t>(1, std::min<size_t>(n_prog, cap / Uc)));
  if (int rc = ensure_state_buffers(h, Uc * Pc, false)) return rc;
  HIPCHK(h->block_cum.reserve((size_t(Uc) * Pc) << (n_eff - 10)));
  if (int rc = values_begin(h, int(Uc * Pc), s)) return rc;  // measurement by-products land in the fixed-point scratch
  if (h->coef_batch_programs < Pc) {
    HIPCHK(h->coef_batch.reserve(size_t(Pc) * stride));
    HIPCHK(launch_replicate(d.coef.p, h->coef_batch.p, uint32_t(d.plan.coef_init.size()), stride, Pc, s));
    h->coef_batch_programs = Pc;
  }
  if (n_eff > uint32_t(kMinTileBits))
    HIPCHK(launch_zero_fill(d_out_counts, ((size_t(n_prog) * size_t(U)) << h->model.n) * sizeof(int32_t), s));
  for (uint32_t s0 = 0; s0 < uint32_t(U); s0 += Uc) {
    const uint32_t c = std::min<uint32_t>(Uc, uint32_t(U) - s0);
    for (uint32_t q0 = 0; q0 < n_prog; q0 += Pc) {
      const uint32_t nq = std::min<uint32_t>(Pc, n_prog - q0);
      HIPCHK(launch_prep_coefs_batch(d.jobs.p, int(d.plan.jobs.size()), d_params, h->coef_batch.p, h->shift_gates.p + q0,
                                     h->shift_vals.p + q0, nq, stride, s));
      HIPCHK(launch_combine_diag(h->coef_batch.p, d.rec_offsets.p, int(d.plan.record_offsets.size()), nq, stride, s));
      for (size_t i = 0; i < d.plan.passes.size(); ++i) {
        const Pass& p = d.plan.passes[i];
        if (p.is_measure_only) continue;
        PassArgs a = d.args[i];
        a.flags = (p.flags & (PASS_INIT_BASIS | PASS_GENERAL | PASS_NO_ZERO_FILL)) | PASS_STORE | PASS_SKIP_MEASURE;
        if (h->opt_force_general) a.flags |= PASS_GENERAL;
        a.prog_states = c;
        a.coef_stride = stride;
        HIPCHK(launch_pass_fwd(p.K, d.plan.R, a, nq * c, h->psi.p, d_bits, h->model.n, d.prog.p, d.tables.p,
                               h->coef_batch.p, h->op_scale.p, h->vals64.p, s0, s));
      }
      HIPCHK(launch_sample_counts(h->psi.p, n_eff, h->model.n, nq * c, c, q0, h->block_cum.p, uint32_t(n_shots), seed, s0,
                                  uint32_t(U), d_out_counts, s));
    }
  }
  return 0;
}

int qhbm_expectation_vjp(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params,
                         const float* d_upstream, float* d_out_vals, float* d_grad, int method,
                         void* stream) {
  if (int rc = check_call(h, U)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int P = h->model.n_params;
  if (U == 0) {
    if (P) HIPCHK(launch_zero_fill(d_grad, size_t(P) * sizeof(float), s));
    return 0;
  }
  const size_t nv = size_t(U) * h->model.n_ops;
  if (!d_out_vals) {
    HIPCHK(h->vals_tmp.reserve(nv));
    d_out_vals = h->vals_tmp.p;
  }
  if (method == QHBM_GRAD_ADJOINT) {
    if (int rc = adjoint_sweep(h, d_bits, U, d_params, d_upstream, d_out_vals, s)) return rc;
    HIPCHK(launch_reduce_grad(h->state_grad.p, uint32_t(U), uint32_t(h->adj.plan.slot_gate.size()),
                              h->param_slot_begin.p, h->param_slots.p, h->slot_factor.p, d_grad, P, 0, s));
    h->state_grad_U = U;
    return 0;
  }
  if (method != QHBM_GRAD_PARAMETER_SHIFT) return fail(h, "unknown gradient method");
  // tfq ParameterShift / baselines/train.py:190-240: exponent c*s -> shift the gate's exponent by
  // +-1/2, weight +-pi*c/2, one pair of forwards per gate occurrence.  The (state, shifted program)
  // pairs are the batch: as many programs as the workspace holds run in ONE launch set, each on
  // its own copy of the coefficient buffer (PassArgs::prog_states).
  if (int rc = forward(h, d_bits, U, d_params, d_out_vals, -1, 0.0, s)) return rc;
  DevicePlan& d = h->fwd;
  const bool from_obs = forward_values_from_observable(h);
  if (!h->shift_ready || h->shift_tables_from_obs != from_obs) {  // shift tables: once per model
    h->shift_tables_from_obs = from_obs;
    std::vector<int> sg, sp;
    std::vector<float> sv, sw;
    for (size_t g = 0; g < h->model.gates.size(); ++g) {
      const Gate& G = h->model.gates[g];
      if (G.param_idx < 0 || G.kind == QHBM_GATE_I || h->model.frozen(G.param_idx)) continue;
      if (G.kind == QHBM_GATE_ISWAPPOW)
        return fail(h, "the two-term parameter-shift rule does not apply to ISWAPPOW; use the adjoint method");
      sg.push_back(int(g)); sg.push_back(int(g));
      sv.push_back(0.5f); sv.push_back(-0.5f);
      sp.push_back(G.param_idx);
      sw.push_back(float(1.5707963267948966 * double(G.scalar)));
    }
    // Execution order: by the first pass that reads the shifted gate's coefficients (first_dependent_pass), gate order
    // within a pass; the sums of program q land at prog_acc[dst[q]], i.e. in GATE order, so the combination per parameter
    // adds in the order it always did.  Without sharing every program "starts" at pass 0.
    const int n_pass = int(d.plan.passes.size());
    int first_measuring = n_pass;
    std::vector<int> first(h->model.gates.size(), 0);
    if (h->opt_shift_prefix) first = first_dependent_pass(d.plan, h->model.gates.size(), &first_measuring);
    auto start_of = [&](int gate) {
      int k = std::min(first[size_t(gate)], n_pass - 1);
      if (!from_obs) k = std::min(k, first_measuring);   // measuring passes write the program's own accumulators
      return std::max(k, 0);
    };
    std::vector<int> order(sg.size());
    for (size_t q = 0; q < order.size(); ++q) order[q] = int(q);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return start_of(sg[size_t(x)]) < start_of(sg[size_t(y)]); });
    std::vector<int> sg_exec(sg.size()), dst(sg.size());
    std::vector<float> sv_exec(sg.size());
    h->shift_group_end.assign(size_t(std::max(n_pass, 1)), 0u);
    for (size_t q = 0; q < order.size(); ++q) {
      sg_exec[q] = sg[size_t(order[q])];
      sv_exec[q] = sv[size_t(order[q])];
      dst[q] = order[q];
      h->shift_group_end[size_t(start_of(sg_exec[q]))] = uint32_t(q + 1);
    }
    for (size_t i = 1; i < h->shift_group_end.size(); ++i)
      h->shift_group_end[i] = std::max(h->shift_group_end[i], h->shift_group_end[i - 1]);
    HIPCHK(hipStreamSynchronize(s));  // synchronous copies into buffers an earlier call on this stream may still read
    HIPCHK(h->shift_gates.upload(sg_exec));
    HIPCHK(h->shift_vals.upload(sv_exec));
    HIPCHK(h->shift_dst.upload(dst));
    HIPCHK(h->shift_param.upload(sp));
    HIPCHK(h->shift_weight.upload(sw));
    h->shift_programs = uint32_t(sg.size());
    h->shift_gate_count = uint32_t(sp.size());
    h->shift_ready = true;
  }
  const uint32_t n_prog = h->shift_programs, n_shift_gates = h->shift_gate_count;
  if (n_prog == 0) {
    if (P) HIPCHK(launch_zero_fill(d_grad, size_t(P) * sizeof(float), s));
    return 0;
  }
  const uint32_t stride = uint32_t((d.plan.coef_init.size() + 64 + 63) / 64 * 64);
  // batch geometry: Uc states x Pc programs per launch set (+ Uc base states when programs share their prefix)
  size_t cap = std::max<size_t>(1, budget_bytes(h) / state_bytes(h));
  uint32_t max_nl = 0;
  for (const PassArgs& a : d.args) max_nl = std::max(max_nl, a.n_nonlocal);
  cap = std::min<size_t>(cap, (size_t(1) << 30) >> max_nl);  // grid.x = elements << n_nonlocal
  cap = std::min<size_t>(cap, (size_t(2) << 30) / (size_t(stride) * sizeof(float)));  // <= 2 GiB of coefficient copies
  if (h->opt_chunk > 0) cap = std::min<size_t>(cap, size_t(h->opt_chunk));
  // sharing needs a second buffer of Uc base states and at least one pass to skip
  const bool share = h->opt_shift_prefix != 0 && h->shift_group_end.size() > 1 && h->shift_group_end[0] < n_prog && cap >= 2;
  const uint32_t Uc = uint32_t(std::min<size_t>(size_t(U), share ? cap / 2 : cap));
  const uint32_t Pc = uint32_t(std::max<size_t>(1, std::min<size_t>(n_prog, (cap - (share ? Uc : 0)) / Uc)));
  if (int rc = ensure_state_buffers(h, Uc * Pc, false)) return rc;
  if (share) HIPCHK(h->lam.reserve(size_t(Uc) << d.plan.n_eff, false));   // the base states live where lambda would
  const size_t nvb = size_t(Uc) * Pc * size_t(h->model.n_ops);
  HIPCHK(h->vals64.reserve(nvb));
  HIPCHK(h->vals_batch.reserve(nvb));
  HIPCHK(h->prog_acc.reserve(n_prog));
  HIPCHK(launch_zero_fill(h->prog_acc.p, size_t(n_prog) * sizeof(double), s));
  if (h->coef_batch_programs < Pc) {
    HIPCHK(h->coef_batch.reserve(size_t(Pc) * stride));
    HIPCHK(launch_replicate(d.coef.p, h->coef_batch.p, uint32_t(d.plan.coef_init.size()), stride, Pc, s));
    h->coef_batch_programs = Pc;
  }
  h->retained_U = 0;
  h->state_grad_U = 0;
  bool measure_only_after = !d.plan.global_terms.empty();
  for (const Pass& p : d.plan.passes) measure_only_after |= p.is_measure_only;
  // The values of a shifted program come the way a forward-only call takes them: where that is the observable kernel
  // (wide terms: lean passes, then one sweep over the final states, the lower block of every pair only), the
  // shifted programs do the same -- config 4's 480 masks cost 51 measurement passes per program otherwise.
  auto pass_args = [&](size_t i, uint32_t prog_states) {
    const Pass& p = d.plan.passes[i];
    PassArgs a = d.args[i];
    a.flags = p.flags & (PASS_INIT_BASIS | PASS_GENERAL | PASS_NO_ZERO_FILL);
    if (h->opt_force_general) a.flags |= PASS_GENERAL;
    if (from_obs) a.flags |= PASS_SKIP_MEASURE;
    if (!p.is_measure_only && (!p.completes_circuit || measure_only_after || from_obs)) a.flags |= PASS_STORE;
    a.prog_states = prog_states;
    a.coef_stride = prog_states ? stride : 0u;
    return a;
  };
  const size_t n_pass = d.plan.passes.size();
  for (uint32_t s0 = 0; s0 < uint32_t(U); s0 += Uc) {
    const uint32_t c = std::min<uint32_t>(Uc, uint32_t(U) - s0);
    // k: the pass the programs of this group start at; the base buffer holds the base program's states after passes < k
    for (size_t k = 0; k < (share ? n_pass : size_t(1)); ++k) {
      const uint32_t group_begin = share ? (k ? h->shift_group_end[k - 1] : 0u) : 0u;
      const uint32_t group_end = share ? h->shift_group_end[k] : n_prog;
      for (uint32_t q0 = group_begin; q0 < group_end; q0 += Pc) {
        const uint32_t nq = std::min<uint32_t>(Pc, group_end - q0);
        HIPCHK(launch_prep_coefs_batch(d.jobs.p, int(d.plan.jobs.size()), d_params, h->coef_batch.p, h->shift_gates.p + q0,
                                       h->shift_vals.p + q0, nq, stride, s));
        HIPCHK(launch_combine_diag(h->coef_batch.p, d.rec_offsets.p, int(d.plan.record_offsets.size()), nq, stride, s));
        HIPCHK(launch_zero_fill(h->vals64.p, size_t(nq) * c * size_t(h->model.n_ops) * sizeof(unsigned long long), s));
        bool first_pass = true;
        for (size_t i = k; i < n_pass; ++i) {
          const Pass& p = d.plan.passes[i];
          if (from_obs && p.is_measure_only) continue;
          const PassArgs a = pass_args(i, c);
          hipEvent_t* ev = timer_begin(h, 0, s);
          // (the first pass a prefix-sharing program runs loads the base program's state of its bitstring)
          HIPCHK(launch_pass_fwd(p.K, d.plan.R, a, nq * c, h->psi.p, d_bits, h->model.n, d.prog.p, d.tables.p,
                                 h->coef_batch.p, h->op_scale.p, h->vals64.p, s0, s,
                                 (first_pass && k > 0) ? h->lam.p : nullptr));
          timer_end(ev, s);
          first_pass = false;
        }
        if (from_obs) {
          if (int rc = run_values_chunk(h, 0u, nq * c, s)) return rc;
        } else if (!d.plan.global_terms.empty()) {
          HIPCHK(launch_measure_global(h->psi.p, uint32_t(d.plan.n_eff), nq * c, h->global_terms.p,
                                       uint32_t(d.plan.global_terms.size()), h->op_scale.p, h->vals64.p,
                                       uint32_t(h->model.n_ops), 0u, s));
        }
        HIPCHK(launch_values_from_fixed(h->vals64.p, h->op_inv_scale.p, h->vals_batch.p,
                                        nq * c * uint32_t(h->model.n_ops), uint32_t(h->model.n_ops), s));
        HIPCHK(launch_shift_program_accumulate(h->vals_batch.p, d_upstream, nq, c, uint32_t(h->model.n_ops), s0,
                                               h->prog_acc.p, s, h->shift_dst.p + q0));
      }
      // the base program advances by pass k (its own coefficients: d.coef, prepared by the forward call above) unless
      // no later program needs it
      if (share && group_end < n_prog && !(from_obs && d.plan.passes[k].is_measure_only)) {
        PassArgs a = pass_args(k, 0u);
        a.flags |= PASS_STORE | PASS_SKIP_MEASURE;   // (its values were taken by the forward call)
        hipEvent_t* ev = timer_begin(h, 0, s);
        HIPCHK(launch_pass_fwd(d.plan.passes[k].K, d.plan.R, a, c, h->lam.p, d_bits, h->model.n, d.prog.p, d.tables.p,
                               d.coef.p, h->op_scale.p, h->vals64.p, s0, s));
        timer_end(ev, s);
      }
    }
  }
  HIPCHK(launch_shift_combine(h->prog_acc.p, h->shift_param.p, h->shift_weight.p, int(n_shift_gates), d_grad, P, s));
  return 0;
}

int qhbm_expectation_jacobian(qhbm_engine* h, const int8_t* d_bits, int U, const float* d_params,
                              float* d_out_vals, float* d_jac, void* stream) {
  if (int rc = check_call(h, U)) return rc;
  if (U == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int T = h->model.n_ops, P = h->model.n_params;
  const size_t nv = size_t(U) * T;
  HIPCHK(h->upstream_tmp.reserve(nv));
  if (!d_out_vals) {
    HIPCHK(h->vals_tmp.reserve(nv));
    d_out_vals = h->vals_tmp.p;
  }
  std::vector<float> up(nv);
  for (int k = 0; k < T; ++k) {
    for (size_t i = 0; i < nv; ++i) up[i] = (int(i % size_t(T)) == k) ? 1.f : 0.f;
    HIPCHK(hipMemcpyAsync(h->upstream_tmp.p, up.data(), nv * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    if (int rc = adjoint_sweep(h, d_bits, U, d_params, h->upstream_tmp.p, d_out_vals, s)) return rc;
    HIPCHK(launch_scatter_jac(h->state_grad.p, uint32_t(U), uint32_t(h->adj.plan.slot_gate.size()),
                              h->param_slot_begin.p, h->param_slots.p, h->slot_factor.p, d_jac, uint32_t(T),
                              uint32_t(k), uint32_t(P), s));
  }
  return 0;
}

int qhbm_num_passes(qhbm_engine* h, int* forward_passes, int* backward_passes) {
  if (!h) return 1;
  if (int rc = build_plans(h)) return rc;
  if (forward_passes) *forward_passes = int(h->fwd.plan.passes.size());
  if (backward_passes) *backward_passes = int(h->adj.plan.passes.size());
  return 0;
}

int qhbm_describe_schedule(qhbm_engine* h, char* buf, size_t buf_len) {
  if (!h || !buf || !buf_len) return 1;
  if (int rc = build_plans(h)) return rc;
  std::string s = describe_plan(h->fwd.plan) + describe_plan(h->adj.plan);
  {  // what the chooser of the adjoint tile and pass order compares (adjoint_plan_seconds)
    char line[96];
    std::snprintf(line, sizeof(line), "adjoint time model: %.2f us per state\n", 1e6 * adjoint_plan_seconds(h->adj.plan, h->model));
    s += line;
  }
  if (!h->model.terms.empty()) {  // which kernel forms lambda = O psi / the values (bench.py names it in `roofline.kernel`)
    s += std::string("observable kernel: lambda = ") + (block_kernel(h) ? "observable_blocks_kernel" : "apply_observable_kernel");
    s += std::string(" values = ") + (gather_multi_mode(h) ? "apply_observable_kernel (an accumulator per observable)"
                                      : multi_value_mode(h) || (value_mode(h) && block_kernel(h)) ? "observable_blocks_kernel"
                                      : value_mode(h)                                            ? "apply_observable_kernel"
                                                                                                  : "measured in the passes");
    s += "\n";
  }
  std::snprintf(buf, buf_len, "%s", s.c_str());
  return 0;
}

int qhbm_kernel_time_ms(qhbm_engine* h, int reset, double* fwd_ms, int64_t* fwd_launches, double* bwd_ms,
                        int64_t* bwd_launches, double* obs_ms, int64_t* obs_launches) {
  if (!h) return 1;
  double ms[3] = {0.0, 0.0, 0.0};
  int64_t cnt[3] = {0, 0, 0};
  if (h->device >= 0 && !h->events.empty()) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->events.back().b));
    for (auto& ev : h->events) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev.a, ev.b) == hipSuccess) { ms[ev.kind] += t; ++cnt[ev.kind]; }
    }
  }
  if (fwd_ms) *fwd_ms = ms[0];
  if (fwd_launches) *fwd_launches = cnt[0];
  if (bwd_ms) *bwd_ms = ms[1];
  if (bwd_launches) *bwd_launches = cnt[1];
  if (obs_ms) *obs_ms = ms[2];
  if (obs_launches) *obs_launches = cnt[2];
  if (reset) {
    h->free_events.insert(h->free_events.end(), h->events.begin(), h->events.end());
    h->events.clear();
  }
  return 0;
}

}  // extern "C"

namespace {

// fp32 operations (FMA = 2) the gate arithmetic of one pass kernel executes per AMPLITUDE of a tile it
// does not skip, from the plan's instance records -- the counts are those of the inline-asm sequences
// in kernels.hip (v_pk_fma_f32 = 4, v_pk_mul_f32 / v_pk_add_f32 = 2 per amplitude pair of lanes):
//   forward   X**t three shears 6 | PH1 3 (half the amplitudes x (mul + fma)) | PH2 1.5 | FULL table 5.625
//             | boundary CPH 3 x (share of waves whose predicate is on) | Y 6 | dense 2x2 14 | dense 4x4 32
//   adjoint   X 16 (psi 6 + lambda 6 + inner product 4) | PH1 8 | PH2 4 | FULL 15.75 (two tables 11.25 + the ten
//             partial sums as scalar differences, round 5: 15 x (mul + fma) + 27 adds = 4.5) | CPH 8 x share | Y 16 |
//             dense 2x2 44 | dense 4x4 on (psi, lambda) + generator 96
// Rounds whose waves are dead (OP_ROUND word 4) run on 2^-popc(dead mask) of the waves.  Wave
// reductions, address arithmetic and record decoding are NOT counted: this is the arithmetic the gate
// set requires of this kernel design, the numerator of a compute roofline against the fp32 vector peak.
double pass_flops_per_amplitude(const Plan& plan, const Pass& p) {
  const RecordLayout L(plan.R, plan.adjoint);
  const bool adj = plan.adjoint;
  double total = 0.0;
  size_t round_i = 0;
  for (size_t pc = 0; pc < p.prog.size();) {
    const uint32_t w0 = p.prog[pc], opc = w0 & 0xffu;
    if (opc == OP_END) break;
    if (opc == OP_ROUND) {
      const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8, first = p.prog[pc + 2], dead = p.prog[pc + 4];
      const uint32_t wmask = round_i < p.round_wavemasks.size() ? p.round_wavemasks[round_i] : 0u;
      const double alive = adj ? 1.0 / double(1u << __builtin_popcount(dead)) : 1.0;
      double f = 0.0;
      for (uint32_t i = 0; i < n_inst; ++i) {
        const uint32_t* rec = &plan.coef_init[first + size_t(i) * size_t(L.words())];
        const uint32_t h0 = rec[0], h1 = rec[1];
        const bool full = (h1 & kFullDiagFlag) != 0;
        f += (adj ? 16.0 : 6.0) * __builtin_popcount(h0 & 0xfu);
        if (full) f += adj ? 15.75 : 5.625;
        else f += (adj ? 8.0 : 3.0) * __builtin_popcount((h0 >> 8) & 0xfu) + (adj ? 4.0 : 1.5) * __builtin_popcount((h0 >> 16) & 0x3fu);
        for (int k = 0; k < 8; ++k) {
          if (!(h1 >> k & 1u)) continue;
          const uint32_t pred = rec[L.pred(k)];
          const bool uniform = (pred >> 8) != 0 || (wmask >> (pred & 0xffu) & 1u);  // tile bit or wave bit: half skip
          f += (adj ? 8.0 : 3.0) * (uniform ? 0.5 : 1.0);
        }
        f += (adj ? 16.0 : 6.0) * __builtin_popcount((h1 >> 16) & 0xfu) + (adj ? 44.0 : 14.0) * __builtin_popcount((h1 >> 24) & 0xfu);
      }
      total += alive * f;
      ++round_i;
      pc += kRoundWords;
    } else if (opc == OP_GATE2) {
      total += adj ? 96.0 : 32.0;
      pc += kGate2Words;
    } else if (opc == OP_MEASURE_WHT) {  // |psi|^2, ten butterfly stages, the terms spread over the threads
      total += 3.0 + 10.0 + 2.0 * double(w0 >> 8) / double(size_t(1) << (plan.K - plan.R));
      pc += size_t(kWhtHeaderWords) + size_t(w0 >> 8) * kMeasTermWords;
    } else {  // OP_MEASURE: [op | n_groups << 8] then groups x {[xl] [n_terms] terms x 4 words}
      const uint32_t n_groups = w0 >> 8;
      ++pc;
      for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t n_terms = p.prog[pc + 1];
        total += 6.0 + double(n_terms);  // conj(psi[l ^ x]) psi[l] (3 FMA) + one signed add per term (upper bound)
        pc += 2 + size_t(n_terms) * kMeasTermWords;
      }
    }
  }
  return total;
}

// Executed micro-ops of one pass, in WAVE-EXECUTIONS per state (a micro-op a wave runs once counts 1): the weights
// of a dynamic instruction mix (scripts/instruction_mix.py multiplies them with the per-micro-op instruction counts of
// the compiled kernel).  Columns: qhbm_engine.h QHBM_CENSUS_*.
void pass_census(const Plan& plan, const Pass& p, double tiles_per_state, double* out) {
  const RecordLayout L(plan.R, plan.adjoint);
  const bool adj = plan.adjoint;
  const double waves = double(size_t(1) << (plan.K - plan.R)) / 64.0 * tiles_per_state;
  out[QHBM_CENSUS_TILES] += tiles_per_state;
  size_t round_i = 0;
  for (size_t pc = 0; pc < p.prog.size();) {
    const uint32_t w0 = p.prog[pc], opc = w0 & 0xffu;
    if (opc == OP_END) break;
    if (opc != OP_ROUND) {
      if (opc == OP_GATE2) pc += kGate2Words;
      else if (opc == OP_MEASURE_WHT) pc += size_t(kWhtHeaderWords) + size_t(w0 >> 8) * kMeasTermWords;
      else {
        const uint32_t n_groups = w0 >> 8;
        ++pc;
        for (uint32_t g = 0; g < n_groups; ++g) pc += 2 + size_t(p.prog[pc + 1]) * kMeasTermWords;
      }
      continue;
    }
    const uint32_t n_inst = (w0 & ~kRoundNoBarrier) >> 8, first = p.prog[pc + 2], dead = p.prog[pc + 4];
    const uint32_t wmask = round_i < p.round_wavemasks.size() ? p.round_wavemasks[round_i] : 0u;
    const double alive = (adj ? 1.0 / double(1u << __builtin_popcount(dead)) : 1.0) * waves;
    out[QHBM_CENSUS_ROUNDS] += waves;
    out[(w0 & kRoundNoBarrier) ? QHBM_CENSUS_ROUNDS_NO_BARRIER : QHBM_CENSUS_ROUNDS_BARRIER] += waves;
    for (uint32_t i = 0; i < n_inst; ++i) {
      const uint32_t* rec = &plan.coef_init[first + size_t(i) * size_t(L.words())];
      const uint32_t h0 = rec[0], h1 = rec[1];
      const bool full = (h1 & kFullDiagFlag) != 0;
      out[QHBM_CENSUS_INSTANCES] += alive;
      for (int j = 0; j < 4; ++j) {
        if (!(h0 >> j & 1u)) continue;
        const bool slot = !adj || rec[L.slot_x(j)] != 0xffffffffu;
        out[slot ? QHBM_CENSUS_X : QHBM_CENSUS_X_NO_SLOT] += alive;
      }
      if (full) out[QHBM_CENSUS_FULL] += alive;
      out[QHBM_CENSUS_PH1] += alive * __builtin_popcount((h0 >> 8) & 0xfu);
      out[QHBM_CENSUS_PH2] += alive * __builtin_popcount((h0 >> 16) & 0x3fu);
      for (int k = 0; k < 8; ++k) {
        if (!(h1 >> k & 1u)) continue;
        const uint32_t pred = rec[L.pred(k)];
        if ((pred >> 8) != 0) { out[QHBM_CENSUS_CPH_TILE_ON] += 0.5 * alive; out[QHBM_CENSUS_CPH_OFF] += 0.5 * alive; }
        else if (wmask >> (pred & 0xffu) & 1u) { out[QHBM_CENSUS_CPH_WAVE_ON] += 0.5 * alive; out[QHBM_CENSUS_CPH_OFF] += 0.5 * alive; }
        else out[QHBM_CENSUS_CPH_LANE] += alive;
      }
      if (adj) {  // eight-wide reductions (instance_adj): CPH group; PH2 group (per-term or FULL with pair terms); X + PH1 group
        if (h1 & 0xffu) out[QHBM_CENSUS_REDUCE8] += alive;
        if (full && ((h0 >> 24) & 0x3fu)) out[QHBM_CENSUS_REDUCE8] += alive;
        if ((h0 >> 16) & 0x3fu) out[QHBM_CENSUS_REDUCE8] += alive;
        if ((h0 & 0xf0fu) || (full && ((h0 >> 4) & 0xfu))) out[QHBM_CENSUS_REDUCE8] += alive;
      }
    }
    ++round_i;
    pc += kRoundWords;
  }
}

}  // namespace

// The chip's sustained packed-fp32 rate and shader clock right now (bench.py calls it straight after its timed region,
// with the chip as warm as the timed kernels left it): one probe launch (kernels.hip clock_probe_kernel) timed with HIP
// events.  ghz = shader cycles per real time inside the waves; cycles_per_pk_fma = kernel time x ghz / instructions of a
// SIMD; tflops = 4 flop x 64 lanes x SIMDs / cycles_per_pk_fma x ghz.  Synchronises the stream.
extern "C" int qhbm_clock_probe(qhbm_engine* h, double* ghz, double* cycles_per_pk_fma, double* tflops, void* stream_v) {
  if (!h) return 1;
  if (int rc = need_device(h)) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, h->device));
  const uint32_t n_cus = uint32_t(prop.multiProcessorCount);
  const uint32_t n_waves = clock_probe_waves(n_cus);
  // (the probe's buffers stay with the engine: bench.py probes after every run)
  DevBuf<uint64_t>& out = h->probe_out;
  DevBuf<float>& sink = h->probe_sink;
  HIPCHK(out.reserve(size_t(2) * n_waves));
  HIPCHK(sink.reserve(1));
  struct Events {  // destroyed on every exit path
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } ev;
  HIPCHK(hipEventCreate(&ev.e0));
  HIPCHK(hipEventCreate(&ev.e1));
  HIPCHK(launch_clock_probe(out.p, sink.p, n_cus, stream));  // warm-up: code object load, clocks
  HIPCHK(hipEventRecord(ev.e0, stream));
  HIPCHK(launch_clock_probe(out.p, sink.p, n_cus, stream));
  HIPCHK(hipEventRecord(ev.e1, stream));
  HIPCHK(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  std::vector<uint64_t> host(size_t(2) * n_waves);
  HIPCHK(hipMemcpy(host.data(), out.p, host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  double cyc = 0.0, ticks = 0.0;
  for (uint32_t w = 0; w < n_waves; ++w) { cyc += double(host[2 * w]); ticks += double(host[2 * w + 1]); }
  if (ticks <= 0.0 || ms <= 0.f) return fail(h, "clock probe: no time elapsed");
  const double g = cyc / ticks * 0.1;  // s_memrealtime ticks at 100 MHz
  const double per = double(ms) * 1e6 * g / clock_probe_instructions_per_simd();
  if (ghz) *ghz = g;
  if (cycles_per_pk_fma) *cycles_per_pk_fma = per;
  if (tflops) *tflops = 256.0 / per * g * 1e9 * double(n_cus) * 4.0 / 1e12;
  return 0;
}

extern "C" int qhbm_plan_builds(qhbm_engine* h, int64_t* forward_plans, int64_t* backward_plans) {
  if (!h) return 1;
  if (forward_plans) *forward_plans = h->fwd_plans_built;
  if (backward_plans) *backward_plans = h->adj_plans_built;
  return 0;
}

extern "C" int qhbm_op_census(qhbm_engine* h, int adjoint, int max_passes, double* out, int* n_passes) {
  if (!h || !out || !n_passes) return 1;
  if (int rc = build_plans(h)) return rc;
  const DevicePlan& d = adjoint ? h->adj : h->fwd;
  std::vector<PassArgs> args;
  std::vector<uint32_t> prog, tables;
  fill_args(d.plan, h->model, &args, &prog, &tables);
  *n_passes = int(args.size());
  for (int i = 0; i < max_passes * QHBM_CENSUS_COLUMNS; ++i) out[i] = 0.0;
  for (size_t i = 0; i < args.size() && int(i) < max_passes; ++i) {
    const Pass& p = d.plan.passes[i];
    // the first forward pass computes on ONE tile per state; later passes on the tiles they launch
    const double tiles = (!adjoint && (p.flags & PASS_INIT_BASIS)) ? 1.0 : double(size_t(1) << args[i].n_free);
    pass_census(d.plan, p, tiles, out + i * QHBM_CENSUS_COLUMNS);
  }
  return 0;
}

namespace {
}  // namespace

extern "C" int qhbm_flop_model(qhbm_engine* h, int U, int with_vjp, double* fwd_flops, double* obs_flops,
                               double* bwd_flops) {
  if (!h) return 1;
  if (int rc = build_plans(h)) return rc;
  const double amps = double(size_t(1) << h->fwd.plan.n_eff) * double(U);
  double f = 0.0, o = 0.0, b = 0.0;
  const bool from_obs = with_vjp ? (value_mode(h) || multi_value_mode(h)) : forward_values_from_observable(h);  // no measurement in the sweep
  {
    std::vector<PassArgs> args;
    std::vector<uint32_t> prog, tables;
    fill_args(h->fwd.plan, h->model, &args, &prog, &tables);
    for (size_t i = 0; i < args.size(); ++i) {
      const Pass& p = h->fwd.plan.passes[i];
      if (from_obs && p.is_measure_only) continue;
      const double live = 1.0 / double(1ull << __builtin_popcount(args[i].zero_mask));
      // the first pass computes on ONE tile per state (the others write zeros and return)
      const double share = (p.flags & PASS_INIT_BASIS) ? 1.0 / double(1ull << p.nonlocal_pos.size()) : live;
      Pass q = p;
      if (from_obs) {  // PASS_SKIP_MEASURE: cut the program at its first measurement
        for (size_t pc = 0; pc < q.prog.size();) {
          const uint32_t opc = q.prog[pc] & 0xffu;
          if (opc == OP_END) break;
          if (opc == OP_MEASURE || opc == OP_MEASURE_WHT) { q.prog[pc] = OP_END; break; }
          pc += opc == OP_ROUND ? size_t(kRoundWords) : size_t(kGate2Words);
        }
      }
      f += share * amps * pass_flops_per_amplitude(h->fwd.plan, q);
    }
  }
  if (with_vjp || from_obs) {
    // lambda = O psi: one packed FMA per amplitude and X-mask group (4), one add per term whose sign varies
    // inside a thread's amplitudes (upper bound: every term), <psi|O|psi> 4 in value mode
    std::vector<uint32_t> xs;
    for (const PauliTerm& t : h->model.terms) xs.push_back(t.x);
    std::sort(xs.begin(), xs.end());
    const double groups = double(std::unique(xs.begin(), xs.end()) - xs.begin());
    o = amps * (4.0 * groups + double(h->model.terms.size()) + (value_mode(h) ? 4.0 : 0.0));
  }
  if (with_vjp) {
    std::vector<PassArgs> args;
    std::vector<uint32_t> prog, tables;
    fill_args(h->adj.plan, h->model, &args, &prog, &tables);
    for (size_t i = 0; i < args.size(); ++i) {
      const double live = 1.0 / double(1ull << __builtin_popcount(args[i].zero_mask));
      b += live * amps * pass_flops_per_amplitude(h->adj.plan, h->adj.plan.passes[i]);
    }
  }
  if (fwd_flops) *fwd_flops = f;
  if (obs_flops) *obs_flops = o;
  if (bwd_flops) *bwd_flops = b;
  return 0;
}

extern "C" {

int qhbm_traffic_model(qhbm_engine* h, int U, int with_vjp, double* fwd_bytes, double* obs_bytes,
                       double* bwd_bytes) {
  if (!h) return 1;
  if (int rc = build_plans(h)) return rc;
  const double tile_all = double(state_bytes(h)) * double(U);  // every tile of every state, once
  double f = 0.0, o = 0.0, b = 0.0;
  bool measure_only_after = !h->fwd.plan.global_terms.empty();
  for (const Pass& p : h->fwd.plan.passes) measure_only_after |= p.is_measure_only;
  const bool from_obs = with_vjp ? (value_mode(h) || multi_value_mode(h)) : forward_values_from_observable(h);  // no measurement in the sweep
  {
    std::vector<PassArgs> fargs;
    std::vector<uint32_t> fprog, ftables;
    fill_args(h->fwd.plan, h->model, &fargs, &fprog, &ftables);
    for (size_t i = 0; i < fargs.size(); ++i) {
      const Pass& p = h->fwd.plan.passes[i];
      if (from_obs && p.is_measure_only) continue;  // the values come from lambda = O psi
      const double live = 1.0 / double(1ull << __builtin_popcount(fargs[i].zero_mask));  // tiles not skipped
      if (!(p.flags & PASS_INIT_BASIS)) f += live * tile_all;  // the first pass writes the basis state, reads nothing
      if (!p.is_measure_only && (!p.completes_circuit || from_obs || with_vjp || measure_only_after)) f += live * tile_all;
    }
  }
  if (!with_vjp && from_obs) o = tile_all;  // psi read (gathered through L2), nothing written
  if (with_vjp) {
    o = 2.0 * tile_all;  // psi read (gathered through L2), lambda written
    if (multi_value_mode(h) && !gather_multi_mode(h)) o += tile_all;  // ... and the launch that returns the values of several observables
    std::vector<PassArgs> args;
    std::vector<uint32_t> prog, tables;
    fill_args(h->adj.plan, h->model, &args, &prog, &tables);
    for (size_t i = 0; i < args.size(); ++i) {
      double live = 1.0 / double(1ull << __builtin_popcount(args[i].zero_mask));  // tiles not skipped
      // a tile without some of the four low index bits still moves whole 128-byte lines
      {
        int low_missing = 0;
        for (uint32_t k = 0; k < args[i].n_nonlocal; ++k) low_missing += args[i].nonlocal_pos[k] < 4;
        live = std::min(1.0, live * double(1u << low_missing));
      }
      const Pass& ap = h->adj.plan.passes[i];
      double write_share = 1.0;  // a relabeling store writes only the amplitudes whose finished bits equal the input
      if (ap.flags & PASS_RELABEL) write_share = 1.0 / double(1u << __builtin_popcount(ap.frozen_new_local));
      b += live * tile_all * 2.0 * (1.0 + ((ap.flags & PASS_STORE) ? write_share : 0.0));
    }
  }
  if (fwd_bytes) *fwd_bytes = f;
  if (obs_bytes) *obs_bytes = o;
  if (bwd_bytes) *bwd_bytes = b;
  return 0;
}

}  // extern "C"
