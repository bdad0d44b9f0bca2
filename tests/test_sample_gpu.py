"""qhbm_sample: computational-basis samples of the final states (tfq.layers.Sample
as used at qnn.py:169,177-181,286-291).  Statistical checks against the oracle's exact
probabilities: every tolerance below is >= 5 standard deviations of the estimator."""
import numpy as np
import pytest
import torch

from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E
from tests.test_engine_gpu import random_circuit

pytestmark = pytest.mark.gpu


def _engine(n, gates, n_params, **opts):
  eng = E.Engine(0)
  for k, v in opts.items():
    eng.set_option(k, v)
  eng.set_circuit(n, gates, n_params)
  return eng


def _index(samples):
  n = samples.shape[-1]
  return (samples.astype(np.int64) * (1 << np.arange(n - 1, -1, -1))).sum(-1)


def test_basis_state_and_reproducibility():
  n = 5
  eng = _engine(n, [(E.GATE_I, 0, -1, -1, 0.0, 1.0)], 0)
  bits = np.array([[1, 0, 1, 1, 0], [0, 0, 0, 0, 0], [1, 1, 1, 1, 1]], np.int8)
  out = eng.sample(bits, np.zeros(0, np.float32), 7, seed=3).cpu().numpy()
  assert out.shape == (3, 7, n) and out.dtype == np.int8
  assert (out == bits[:, None, :]).all()          # no gate: every shot returns the input
  gates, names = O.hea_gates(n, 2, "s")
  params = np.random.default_rng(0).uniform(-1, 1, len(names)).astype(np.float32)
  eng = _engine(n, gates, len(names))
  a = eng.sample(bits, params, 64, seed=11).cpu().numpy()
  b = eng.sample(bits, params, 64, seed=11).cpu().numpy()
  c = eng.sample(bits, params, 64, seed=12).cpu().numpy()
  assert (a == b).all() and (a != c).any()
  # shots of a state do not depend on which other states share the call (counter = (state row, shot))
  d = eng.sample(bits[:1], params, 64, seed=11).cpu().numpy()
  assert (d[0] == a[0]).all()
  assert eng.sample(bits[:0], params, 5).shape == (0, 5, n)
  assert eng.sample(bits, params, 0).shape == (3, 0, n)


@pytest.mark.parametrize("n,tile", [(3, 0), (10, 0), (12, 10)])
def test_distribution_matches_born_rule(n, tile):
  rng = np.random.default_rng(100 + n)
  n_params = 5
  gates = random_circuit(rng, n, 30, n_params)
  params = rng.uniform(-1, 1, n_params)
  bits = rng.integers(0, 2, size=(2, n)).astype(np.int8)
  opts = {"tile_qubits": tile} if tile else {}
  eng = _engine(n, gates, n_params, **opts)
  shots = 200000
  samples = eng.sample(bits, params, shots, seed=2024).cpu().numpy()
  for row in range(2):
    probs = np.abs(O.simulate(n, gates, params, list(bits[row])).ravel())**2
    counts = np.bincount(_index(samples[row]), minlength=2**n)
    assert counts[probs < 1e-12].sum() == 0
    # per-qubit marginals: |p_hat - p| <= 5 sqrt(p(1-p)/shots)
    for q in range(n):
      p1 = probs.reshape((2,) * n).sum(axis=tuple(a for a in range(n) if a != q))[1]
      got = samples[row, :, q].mean()
      assert abs(got - p1) <= 5 * np.sqrt(max(p1 * (1 - p1), 1e-6) / shots) + 1e-4
    # the 16 likeliest outcomes individually, and the total variation on a coarse 16-bin histogram
    top = np.argsort(probs)[-16:]
    for k in top:
      assert abs(counts[k] / shots - probs[k]) <= 5 * np.sqrt(probs[k] * (1 - probs[k]) / shots) + 1e-5
    coarse_p = probs.reshape(16, -1).sum(1) if n >= 4 else probs
    coarse_c = counts.reshape(16, -1).sum(1) / shots if n >= 4 else counts / shots
    assert np.abs(coarse_p - coarse_c).sum() < 0.02


def test_shifted_program():
  """One program of ParameterShift.get_gradient_circuits: gate g's exponent moved by +-1/2."""
  n = 4
  gates, names = O.hea_gates(n, 1, "p")
  params = np.random.default_rng(5).uniform(-1, 1, len(names))
  eng = _engine(n, gates, len(names))
  bits = np.zeros((1, n), np.int8)
  shots = 100000
  g = 0  # X**sx on qubit 0
  samples = eng.sample(bits, params, shots, seed=9, shift_gate=g, shift=0.5).cpu().numpy()
  shifted = list(gates)
  k, q0, q1, p, s, o = shifted[g]
  shifted[g] = (k, q0, q1, p, s, o + 0.5)
  probs = np.abs(O.simulate(n, shifted, params, [0] * n).ravel())**2
  base = np.abs(O.simulate(n, gates, params, [0] * n).ravel())**2
  p1 = probs.reshape((2,) * n).sum(axis=(1, 2, 3))[1]
  b1 = base.reshape((2,) * n).sum(axis=(1, 2, 3))[1]
  got = samples[0, :, 0].mean()
  assert abs(p1 - b1) > 0.05                      # the shift matters for this marginal
  assert abs(got - p1) <= 5 * np.sqrt(p1 * (1 - p1) / shots) + 1e-4
  with pytest.raises(E.EngineError, match="shift_gate"):
    eng.sample(bits, params, 4, shift_gate=len(gates))


@pytest.mark.parametrize("n,tile,shots", [(3, 0, 400000), (10, 0, 200000), (12, 10, 60000)])
def test_sample_counts_of_shifted_programs_match_the_born_rule(n, tile, shots):
  """qhbm_sample_counts: per-outcome shot counts of several shifted programs in ONE launch set (the
  2 G programs of tfq's ParameterShift on sampled estimates, qnn.py:188-226).  n <= 10 takes the
  thread-per-shot kernel, n = 12 the wave-per-shot one with global counters.  Every count is checked
  against the oracle's probabilities of the program it belongs to (5 sigma), rows add up to the shot
  count, and a call is reproducible and independent of how the batch is cut."""
  rng = np.random.default_rng(300 + n)
  gates, names = O.hea_gates(n, 2, "c")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  opts = {"tile_qubits": tile} if tile else {}
  eng = _engine(n, gates, len(names), **opts)
  programs = [(-1, 0.0), (0, 0.5), (0, -0.5), (len(gates) - 1, 0.5), (3, -0.5)]
  sg, sv = [g for g, _ in programs], [s for _, s in programs]
  counts = eng.sample_counts(bits, params, shots, seed=77, shift_gates=sg, shifts=sv)
  assert counts.shape == (5, 3, 2**n) and counts.dtype == torch.int32
  counts = counts.cpu().numpy()
  assert (counts.sum(-1) == shots).all() and (counts >= 0).all()
  for q, (g, sh) in enumerate(programs):
    shifted = list(gates)
    if g >= 0:
      k, q0, q1, p, s, o = shifted[g]
      shifted[g] = (k, q0, q1, p, s, o + sh)
    for u in range(3):
      probs = np.abs(O.simulate(n, shifted, params, list(bits[u])).ravel())**2
      assert counts[q, u][probs < 1e-12].sum() == 0
      top = np.argsort(probs)[-12:]
      err = np.abs(counts[q, u][top] / shots - probs[top])
      assert (err <= 5 * np.sqrt(probs[top] * (1 - probs[top]) / shots) + 1e-5).all(), (q, u, err.max())
      coarse = 8 if n >= 3 else 2**n
      tv = np.abs(probs.reshape(coarse, -1).sum(1) - counts[q, u].reshape(coarse, -1).sum(1) / shots).sum()
      assert tv < 0.02, (q, u, tv)
  # the shifted programs really differ from the unshifted one
  assert np.abs(counts[1, 0] - counts[0, 0]).sum() > 0.02 * shots
  # reproducible; independent of the batch composition and of the launch-set geometry
  again = eng.sample_counts(bits, params, shots, seed=77, shift_gates=sg, shifts=sv).cpu().numpy()
  np.testing.assert_array_equal(again, counts)
  other = eng.sample_counts(bits, params, shots, seed=78, shift_gates=sg, shifts=sv).cpu().numpy()
  assert (other != counts).any()
  # shots are keyed by (seed, program position, state row, shot): the first state alone gives the same counts
  one = eng.sample_counts(bits[:1], params, shots, seed=77, shift_gates=sg[:2], shifts=sv[:2]).cpu().numpy()
  np.testing.assert_array_equal(one[:, 0], counts[:2, 0])
  cut = _engine(n, gates, len(names), chunk_states=2, **opts)
  np.testing.assert_array_equal(cut.sample_counts(bits, params, shots, seed=77, shift_gates=sg, shifts=sv).cpu().numpy(),
                                counts)
  # degenerate calls
  assert eng.sample_counts(bits[:0], params, 10, shift_gates=sg, shifts=sv).shape == (5, 0, 2**n)
  assert int(eng.sample_counts(bits, params, 0, shift_gates=[-1], shifts=[0.0]).sum()) == 0
  with pytest.raises(E.EngineError, match="out of range"):
    eng.sample_counts(bits, params, 4, shift_gates=[len(gates)], shifts=[0.5])


def test_negative_shift_gates_mean_unshifted_and_constant_gates_cannot_be_shifted():
  """any shift_gate < 0 is the unshifted circuit -- also -2, the scheduler's internal tag of the fixed ops a
  gate is lowered to (a Y power = S . X**t . S^dagger: its two S phases must never be shifted) -- and a gate with a
  constant exponent has no shifted program (its lowering would apply the shift to the wrong ops): rejected."""
  n = 3
  gates = [(O.GATE_HPOW, 0, -1, -1, 0.0, 1.0),        # constant H: lowered to three fixed ops
           (O.GATE_YPOW, 1, -1, 0, 1.0, 0.0),         # Y**t: S X**t S^dagger
           (O.GATE_CNOTPOW, 0, 2, -1, 0.0, 1.0)]
  eng = E.Engine(0)
  eng.set_circuit(n, gates, 1)
  bits = np.zeros((1, n), np.int8)
  params = np.array([0.3], np.float32)
  base = eng.sample(bits, params, 4096, seed=5)
  for g in (-1, -2, -7):
    assert torch.equal(eng.sample(bits, params, 4096, seed=5, shift_gate=g, shift=0.5), base)
  c0 = eng.sample_counts(bits, params, 4096, seed=9, shift_gates=[-1, -2], shifts=[0.0, 0.5])
  np.testing.assert_array_equal(c0[0].cpu().numpy(), eng.sample_counts(bits, params, 4096, seed=9)[0].cpu().numpy())
  # same distribution for the "-2" program (its own random stream: compare frequencies, not shots)
  f = c0.float().cpu().numpy() / 4096
  assert np.abs(f[0] - f[1]).max() < 0.05
  with pytest.raises(E.EngineError, match="constant exponent"):
    eng.sample(bits, params, 16, seed=1, shift_gate=0, shift=0.5)
  with pytest.raises(E.EngineError, match="constant exponent"):
    eng.sample_counts(bits, params, 16, seed=1, shift_gates=[1, 2], shifts=[0.5, -0.5])
  # the parametrised gate shifts as before
  shifted = eng.sample_counts(bits, params, 1 << 14, seed=2, shift_gates=[1], shifts=[0.5])[0, 0].float().cpu().numpy() / (1 << 14)
  want = np.abs(O.simulate(n, gates[:1] + [(O.GATE_YPOW, 1, -1, -1, 0.0, 0.8)] + gates[2:], np.zeros(0), bits[0]).reshape(-1)) ** 2
  assert np.abs(shifted - want).max() < 0.03
