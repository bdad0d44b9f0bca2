"""The C-ABI shared library: loads on a CPU-only box, exports every symbol the
header declares, keeps the enum values the oracle and host use, and its
planning side (scheduler) works without a device.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qhbm_engine.h")

pytestmark = pytest.mark.skipif(not os.path.exists(E.LIB_PATH),
                                reason="engine library not built (run __graft_entry__.build())")


def _header_text():
  with open(HEADER) as f:
    return f.read()


def test_every_declared_symbol_is_exported():
  text = re.sub(r"/\*.*?\*/", "", _header_text(), flags=re.S)
  declared = set(re.findall(r"\b(qhbm_[a-z_]+)\s*\(", text))
  assert declared == set(E.ABI_SYMBOLS)
  lib = ctypes.CDLL(E.LIB_PATH)
  for sym in declared:
    assert hasattr(lib, sym), sym
  assert lib.qhbm_abi_version() == 5


def test_gate_kind_enum_matches_host_and_oracle():
  text = _header_text()
  for name, value in re.findall(r"QHBM_GATE_([A-Z]+)\s*=\s*(\d+)", text):
    if name == "KIND_COUNT":
      continue
    assert getattr(E, f"GATE_{name}") == int(value)
    assert getattr(O, f"GATE_{name}") == int(value)
  # struct qhbm_gate, ABI v3: four int32, scalar, offset, global_shift
  assert ctypes.sizeof(E.QhbmGate) == 28
  fields = re.search(r"typedef struct qhbm_gate \{(.*?)\} qhbm_gate;", text, re.S).group(1)
  assert re.findall(r"(?:int32_t|float) (\w+);", fields) == [f[0] for f in E.QhbmGate._fields_]


def _planner(n, layers, op, **opts):
  eng = E.Engine(device=None)
  for k, v in opts.items():
    eng.set_option(k, v)
  gates, names = O.hea_gates(n, layers)
  eng.set_circuit(n, gates, len(names))
  eng.set_observables([op])
  return eng


def test_planning_without_device_and_loud_compute_failure():
  eng = _planner(4, 2, O.tfim_ring_op(4))
  assert eng.num_passes() == (1, 1)
  assert "forward plan" in eng.describe_schedule()
  with pytest.raises(E.EngineError, match="no CPU fallback|no device"):
    eng.expectation(np.zeros((1, 4), np.int8), np.zeros(22, np.float32))


@pytest.mark.parametrize("n,layers,max_fwd,max_bwd,max_bwd_relabel",
                         [(12, 8, 1, 1, 1), (20, 16, 7, 6, 9), (24, 16, 10, 9, 10), (28, 32, 16, 21, 26)])
def test_baseline_configs_schedule(n, layers, max_fwd, max_bwd, max_bwd_relabel):
  """Light-cone scheduling: far fewer HBM passes than gates (944 gates at n=20).  The bounds are the
  pass counts of the scheduler whose diagonal terms wait only for non-diagonal gates (they commute
  with each other): with every pending op a barrier the chain's cone shrinks twice as fast and config
  3 needs 10 + 11 passes instead of 7 + 6.  The default (relabeling) adjoint plans take a few more,
  smaller tail passes -- a pruned pass moves only its live lines there -- and the engine keeps the candidate order
  with the least modelled time, which for deep circuits is one with more, better-pruned passes (config 5: 26, or
  18 with tiles of 2^13 amplitudes -- a tie on the hardware; `adjoint_plan_search` = 0: the scheduler's first choice)."""
  op = O.xxz_chain_op(n) if n == 20 else O.tfim_ring_op(n)
  eng = _planner(n, layers, op, adjoint_relabel=0)
  fwd, bwd = eng.num_passes()
  assert 1 <= fwd <= max_fwd
  assert 1 <= bwd <= max_bwd
  assert eng.workspace_bytes(8) >= 8 * 8 * 2**n or n > 24
  fwd_r, bwd_r = _planner(n, layers, op).num_passes()
  assert fwd_r == fwd and 1 <= bwd_r <= max_bwd_relabel
  assert _planner(n, layers, op, adjoint_plan_search=0).num_passes()[1] <= min(bwd_r, 20)


def test_first_tile_of_a_chain_absorbs_the_whole_triangle():
  """12 local qubits at the end of a nearest-neighbour chain: 12 + 11 + ... + 1 = 78 one-qubit gates
  are inside the light cone of a first pass (the forward plan's second pass adds them to the 30 of its
  first: 110 with the next layers; the plain adjoint layout starts with exactly the 78)."""
  eng = _planner(20, 16, O.xxz_chain_op(20), cph_wave_bits=0)
  text = eng.describe_schedule()
  adjoint = text[text.index("adjoint plan"):]
  first = [line for line in adjoint.splitlines() if line.strip().startswith("pass 0:")][0]
  assert "mat_ops=78" in first, first


def test_adjoint_pass_order_is_searched_for_early_finished_bits():
  """The adjoint plan orders its passes so that the low index bits run out of gates early (the 136-gate
  triangle 16 + 15 + ... + 1 finishes the first qubit of the chain): config 3 runs at most 150 of its 320
  one-qubit gates before the first pruned pass -- the greedy order needs 220.  Without relabeling the
  later tiles drop every finished low bit (c = 0) and prune on it; the default, relabeling plan moves a
  finished bit out of the 128-byte lines when the finishing pass stores (`moves-local-bits`), keeps every
  tile made of whole lines (c >= 4), and its byte model is well below the other's."""
  import re
  eng = _planner(20, 16, O.xxz_chain_op(20), adjoint_relabel=0)
  text = eng.describe_schedule()
  adjoint = text[text.index("adjoint plan"):]
  mats = [int(x) for x in re.findall(r"mat_ops=(\d+)", adjoint)]
  cs = [int(x) for x in re.findall(r" c=(\d+) ", adjoint)]
  assert sum(mats) == 320 and len(mats) <= 7
  first_c0 = cs.index(0)
  assert 136 <= sum(mats[:first_c0]) <= 150
  assert sum(m for m, c in zip(mats, cs) if c == 0) >= 150
  rel = _planner(20, 16, O.xxz_chain_op(20))
  text = rel.describe_schedule()
  adjoint = text[text.index("adjoint (relabeling) plan"):]
  lines = [ln for ln in adjoint.splitlines() if ln.strip().startswith("pass ")]
  mats = [int(re.search(r"mat_ops=(\d+)", ln).group(1)) for ln in lines]
  assert sum(mats) == 320 and all(int(re.search(r" c=(\d+) ", ln).group(1)) >= 4 for ln in lines)
  first_move = next(i for i, ln in enumerate(lines) if "moves-local-bits=" in ln)
  assert 136 <= sum(mats[:first_move + 1]) <= 150          # the pass that finishes the first bit ends the unpruned part
  assert rel.traffic_model(64, True)["bwd_bytes"] < 0.7 * eng.traffic_model(64, True)["bwd_bytes"]
  assert rel.flop_model(64, True)["bwd_flops"] <= 1.02 * eng.flop_model(64, True)["bwd_flops"]


def test_adjoint_plan_is_chosen_by_the_time_model():
  """The backward plan is the one with the least modelled time (per pass: arithmetic of the flop model at the rate
  the kernel sustains, or tile traffic) among the scheduler's best pass orders and the greedy one, with tiles of
  2^12 and -- `adjoint_tile_qubits` = 0 -- of 2^13 amplitudes (scripts/experiments/adj_tile_ab.sh, adj_search_ab.sh): config 3
  keeps the scheduler's first choice on 2^12, deep TFIM circuits (nothing finishes early) take other orders and
  / or the larger tile.  Explicit options are obeyed."""
  def tile_bits(eng):
    text = eng.describe_schedule()
    return int(re.search(r"tile_bits=(\d+)", text[text.index("adjoint"):]).group(1))
  c3 = _planner(20, 16, O.xxz_chain_op(20))
  first = _planner(20, 16, O.xxz_chain_op(20), adjoint_plan_search=0)
  assert tile_bits(c3) == 12 and c3.describe_schedule() == first.describe_schedule()
  deep_first = _planner(24, 32, O.tfim_ring_op(24), adjoint_plan_search=0)
  assert tile_bits(deep_first) == 13
  forced = _planner(24, 32, O.tfim_ring_op(24), adjoint_tile_qubits=12, adjoint_plan_search=0)
  assert tile_bits(forced) == 12
  assert deep_first.flop_model(1, True)["bwd_flops"] < 0.98 * forced.flop_model(1, True)["bwd_flops"]
  deep = _planner(24, 32, O.tfim_ring_op(24))              # another order: more passes, less arithmetic
  assert deep.num_passes()[1] > deep_first.num_passes()[1]
  assert deep.flop_model(1, True)["bwd_flops"] < 0.97 * deep_first.flop_model(1, True)["bwd_flops"]
  assert tile_bits(_planner(26, 32, O.tfim_ring_op(26))) == 13
  assert tile_bits(_planner(20, 16, O.xxz_chain_op(20), adjoint_tile_qubits=13)) == 13


def test_schedule_options_and_errors():
  eng = _planner(14, 2, O.xxz_chain_op(14), tile_qubits=10, adjoint_tile_qubits=10)
  fwd, bwd = eng.num_passes()
  assert fwd > 1 and bwd > 1
  with pytest.raises(E.EngineError):
    E.Engine(device=None).set_option("no_such_option", 1)
  with pytest.raises(E.EngineError):
    E.Engine(device=None).set_option("observable_block_bits", 11)   # 12 or 13 (include/qhbm_engine.h)
  E.Engine(device=None).set_option("observable_block_bits", 12)
  bad = E.Engine(device=None)
  with pytest.raises(E.EngineError):
    bad.set_circuit(3, [(O.GATE_CZPOW, 0, 0, -1, 0.0, 1.0)], 0)  # q1 == q0
  with pytest.raises(E.EngineError):
    bad.set_circuit(40, [], 0)
  bad.set_circuit(3, [], 0)
  with pytest.raises(E.EngineError):
    bad.set_observables([[(1.0, 8, 0)]])
  # every gate kind schedules
  rng = np.random.default_rng(0)
  gates = []
  for kind in range(12):
    q0 = int(rng.integers(12))
    q1 = (q0 + 1 + int(rng.integers(11))) % 12 if O.gate_num_qubits(kind) == 2 else -1
    gates.append((kind, q0, q1, kind % 3, 0.5, 0.1))
  eng = E.Engine(device=None)
  eng.set_option("tile_qubits", 10)
  eng.set_circuit(12, gates, 3)
  eng.set_observables([O.tfim_ring_op(12)])
  assert eng.num_passes()[0] >= 1


def test_flop_model_counts_the_gate_arithmetic_of_the_plan():
  """qhbm_flop_model on a planning-only engine: a 12-qubit depth-1 HEA runs in one tile and one pass, so
  the count can be written down: per amplitude 6 flop per X**t (three packed shears), and the 12 Z**t +
  11 CZ**t phases at between 1.5 (PH2) and 5.625 (one FULL table for several) each."""
  n = 12
  eng = _planner(n, 1, O.tfim_ring_op(n))
  fm = eng.flop_model(10, with_vjp=False)
  per_amp = fm["fwd_flops"] / (10 * 2**n)
  meas = sum(6.0 + 1.0 * k for k in (n,) + (1,) * n)    # the ZZ group (x = 0, n terms) and n single-X groups: upper bound
  assert 6.0 * n + 1.5 * 23 * 0.5 <= per_amp <= 6.0 * n + 5.625 * 23 + meas
  assert fm["obs_flops"] == 0.0 and fm["bwd_flops"] == 0.0
  both = eng.flop_model(10, with_vjp=True)
  # (adjoint rounds late in the pass run on the waves that are not dead only: a loose lower bound)
  assert 16.0 * n * 10 * 2**n / 4 <= both["bwd_flops"] <= (16.0 * n + 17.25 * 23) * 10 * 2**n and both["obs_flops"] > 0
  # linear in the batch
  assert eng.flop_model(20, with_vjp=True)["bwd_flops"] == pytest.approx(2 * both["bwd_flops"])
  # config 3: zero tiles and dead waves are excluded -- well below the unpruned count, above the X gates alone / 4
  eng3 = _planner(20, 16, O.xxz_chain_op(20))
  f3 = eng3.flop_model(1, with_vjp=True)
  assert 16.0 * 320 * 2**20 / 4 < f3["bwd_flops"] < (16.0 * 320 + 17.25 * 400) * 2**20


def test_op_census_counts_the_micro_ops_the_plan_executes():
  """qhbm_op_census (ABI v5): executed micro-ops per pass in wave-executions per state -- the weights of the dynamic
  instruction mix (scripts/instruction_mix.py).  Config 3's circuit: every one of the 320 X gates runs on every live
  wave; the census agrees with the flop model's count of X work; planning-only engines answer; so does
  qhbm_plan_builds."""
  n, layers = 20, 16
  eng = _planner(n, layers, O.xxz_chain_op(n))
  fwd, adj = eng.op_census(adjoint=False), eng.op_census(adjoint=True)
  assert (len(fwd), len(adj)) == eng.num_passes()
  for row in fwd + adj:
    assert set(row) == set(E.Engine.CENSUS_COLUMNS) and all(np.isfinite(v) and v >= 0 for v in row.values())
  assert sum(r["reduce8"] for r in fwd) == 0 and sum(r["reduce8"] for r in adj) > 0
  # the second forward pass works on the 32 tiles the first one can have reached; full passes on all 256
  assert fwd[0]["tiles"] == 1 and max(r["tiles"] for r in fwd) == 256
  # an un-pruned wave executes each of the 320 X gates once: 64 waves per tile-set of a state at most
  waves_full = 256 * 4
  x_total = sum(r["x"] + r["x_no_slot"] for r in adj)
  assert 320 * 0.3 * waves_full < x_total <= 320 * waves_full
  assert sum(r["x_no_slot"] for r in adj) == 0                      # every parameter trainable
  frozen = _planner(n, layers, O.xxz_chain_op(n))
  frozen.set_gradient_mask(np.arange(frozen.n_params) % 2 == 0)
  assert sum(r["x_no_slot"] for r in frozen.op_census(adjoint=True)) > 0
  assert eng.plan_builds() == (1, 1) and frozen.plan_builds()[1] >= 1
  with pytest.raises(E.EngineError):
    eng.clock_probe()                                               # needs the device: no CPU fallback


def test_graft_entry_build_checks_the_header_version():
  """__graft_entry__.build() compares the library's ABI version with the header's (a hard-coded number went stale
  when the ABI moved to v4); its source must not pin a literal."""
  with open(os.path.join(ROOT, "__graft_entry__.py")) as f:
    text = f.read()
  assert "QHBM_ABI_VERSION" in text and not re.search(r"qhbm_abi_version\(\) == \d", text)
  declared = int(re.search(r"#define QHBM_ABI_VERSION (\d+)", _header_text()).group(1))
  assert ctypes.CDLL(E.LIB_PATH).qhbm_abi_version() == declared


def test_schedule_names_the_observable_kernel_of_the_operator():
  """`qhbm_describe_schedule` ends with the kernel that forms lambda = O psi and the values for the installed operator
  (bench.py quotes it as `roofline.kernel`): the gather kernel for a chain Hamiltonian, the block-grouped kernel for
  hundreds of X-masks (engine.cpp block_kernel), the passes themselves for sums of single flips and diagonal terms."""
  last = lambda eng: eng.describe_schedule().splitlines()[-1]
  assert last(_planner(20, 16, O.xxz_chain_op(20))) == "observable kernel: lambda = apply_observable_kernel values = apply_observable_kernel"
  many = _planner(16, 2, O.random_pauli_op(16, 300, 7, p_identity=0.75))
  assert last(many) == "observable kernel: lambda = observable_blocks_kernel values = observable_blocks_kernel"
  small = _planner(6, 2, O.tfim_ring_op(6))
  assert last(small).startswith("observable kernel: lambda = apply_observable_kernel")
