"""`inference.CapturedLoss`: a whole VQT / QMHL step -- loss and backward -- recorded into ONE hipGraph and replayed.

The replay runs the same kernels on the same inputs as the eager step over the same padded multiset, so it must return the
SAME BITS (loss and every gradient, atol = 0) on EVERY replay -- round 6 found the second and later replays wrong while the
first was right (memset / memcpy nodes of a graph replayed on another stream than it was warmed up on; 18 qubits and more
showed it in the circuit gradient); against the plain eager step over the unpadded multiset it may differ by the
summation order of the sample averages only (<= 1e-6).  Reference of the step: /root/reference/qhbmlib/inference/
vqt_loss.py:25-55, qmhl_loss.py:21-34, ebm.py:262-329 (the sample average and its score-function gradient).
"""
import numpy as np
import pytest
import torch

from qhbmlib_amd import _engine as E
from qhbmlib_amd import data, inference, ir, models, utils
from tests.test_host_api import hea_circuit

pytestmark = pytest.mark.gpu


def _tfim(qubits):
  ham = ir.PauliSum()
  for i, q in enumerate(qubits):
    ham += -1.0 * ir.PX(q)
    ham += -1.0 * ir.PZ(q) * ir.PZ(qubits[(i + 1) % len(qubits)])
  return ham


def _model(n, layers, samples, kind, seed, name):
  qubits = ir.GridQubit.rect(1, n)
  torch.manual_seed(seed)
  circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, name), tfq_compat_bit_order=False).to("cuda")
  energy = (models.BernoulliEnergy(list(range(n))) if kind == "bernoulli" else models.KOBE(list(range(n)), 2)).to("cuda")
  with torch.no_grad():
    circuit.trainable_variables[0].uniform_(-1, 1)
    energy.post_process[0].kernel.uniform_(-0.4, 0.4)
  e_inf = (inference.BernoulliEnergyInference if kind == "bernoulli" else inference.AnalyticEnergyInference)(
      energy, samples, initial_seed=seed)
  qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit))
  return qubits, qhbm, list(energy.parameters()) + circuit.trainable_variables


def _multiset(e_inf, num):
  with torch.no_grad():
    drawn = e_inf.sample(num).cuda()
  rows, _, counts = utils.unique_bitstrings_with_counts(drawn)
  return rows, counts


def _grads(variables):
  return [v.grad.detach().clone() for v in variables]


@pytest.mark.parametrize("n,layers,samples,kind", [(4, 2, 32, "bernoulli"), (12, 3, 256, "bernoulli"), (10, 2, 128, "kobe"),
                                                   (18, 6, 1024, "kobe")])
def test_replayed_vqt_step_returns_the_bits_of_the_eager_step(n, layers, samples, kind):
  qubits, qhbm, variables = _model(n, layers, samples, kind, 11, "cv")
  e_inf, ham = qhbm.e_inference, _tfim(qubits)
  loss_fn = lambda: inference.vqt(qhbm, [ham], 0.7)
  step = inference.CapturedLoss(loss_fn, [e_inf], variables)
  ms_a, ms_b = _multiset(e_inf, samples), _multiset(e_inf, samples // 2)          # different numbers of unique rows
  for multiset in (ms_a, ms_b, ms_a):
    want_loss = step.eager([multiset]).clone()
    want = _grads(variables)
    got_loss = step([multiset])
    torch.cuda.synchronize()
    assert step.captured and torch.equal(got_loss, want_loss)
    for v, w in zip(variables, want):
      assert torch.equal(v.grad, w)
    # ... and the plain mirror step over the unpadded multiset agrees to rounding
    for v in variables:
      v.grad = None
    with e_inf.fixed_samples(*multiset):
      plain = loss_fn()
      plain.backward()
    assert abs(float(plain) - float(want_loss)) <= 2e-6 * max(1.0, abs(float(plain)))
    for v, w in zip(variables, want):
      np.testing.assert_allclose(v.grad.cpu().numpy(), w.cpu().numpy(), atol=2e-6 * max(1.0, float(w.abs().max())))
  # the variables may be updated in place between replays (an optimiser step): the graph reads the new values
  with torch.no_grad():
    for v in variables:
      v.add_(0.05 * torch.randn_like(v))
  want_loss = step.eager([ms_b]).clone()
  want = _grads(variables)
  got_loss = step([ms_b])
  torch.cuda.synchronize()
  assert torch.equal(got_loss, want_loss) and all(torch.equal(v.grad, w) for v, w in zip(variables, want))
  # drawing its own samples: the sampler runs outside the graph, the step stays finite and close to the exact loss
  loss = step()
  torch.cuda.synchronize()
  assert torch.isfinite(loss) and 1 <= step.last_unique_rows[0] <= samples


@pytest.mark.parametrize("synchronize", [True, False])
def test_thirty_replays_interleaved_with_other_work_on_the_caller_s_stream(synchronize):
  """Round 6: with the log partition of an analytic EBM taken by ONE `logsumexp` over its 2^n energies (a multi-block
  reduction: torch zeroes its semaphores with a memset, a memset NODE under capture), a graph whose replays interleaved with
  unrelated kernels on the caller's stream returned a wrong log Z after about ten steps and stayed wrong.  The analytic
  inference reduces in two single-block stages now (`ebm._logsumexp_rows`), and `CapturedLoss(synchronize=True)` waits for
  every replay; both settings must hold the eager bits through thirty such steps."""
  n, samples = 18, 512
  qubits, qhbm, variables = _model(n, 3, samples, "kobe", 5, "il")
  e_inf, ham = qhbm.e_inference, _tfim(qubits)
  step = inference.CapturedLoss(lambda: inference.vqt(qhbm, [ham], 1.0), [e_inf], variables, synchronize=synchronize)
  multiset = _multiset(e_inf, samples)
  want_loss = step.eager([multiset]).clone()
  want = _grads(variables)
  other = torch.zeros(4096, device="cuda")
  for _ in range(30):
    got = step([multiset])
    other.add_(1.0)                                     # (the caller's own work, never waited for)
  torch.cuda.synchronize()
  assert float(other[0]) == 30.0
  assert torch.equal(got, want_loss) and all(torch.equal(v.grad, w) for v, w in zip(variables, want))
  # the same two-stage reductions outside a graph: log Z and entropy against the one-shot forms
  energies = e_inf.all_energies.detach()
  np.testing.assert_allclose(float(e_inf.log_partition()), float(torch.logsumexp(-energies.double(), 0)), rtol=1e-6)
  logp = -energies.double() - torch.logsumexp(-energies.double(), 0)
  np.testing.assert_allclose(float(e_inf.entropy()), float(-(logp.exp() * logp).sum()), rtol=1e-5)


def test_replayed_qmhl_step_with_two_sample_averages():
  """qmhl(data, model) = <K_model>_data + log Z_model (qmhl_loss.py:33-34): the data QHBM's sample average over ITS
  multiset, the model's modular Hamiltonian measured behind U_data U_model^dagger, gradients for the model only."""
  n, samples = 6, 64
  qubits, model, model_vars = _model(n, 2, samples, "kobe", 3, "qm")
  _, data_qhbm, _ = _model(n, 1, samples, "bernoulli", 4, "qd")
  for p in data_qhbm.parameters():
    p.requires_grad_(False)               # a fixed data source
  source = data.QHBMData(data_qhbm)
  loss_fn = lambda: inference.qmhl(source, model)
  step = inference.CapturedLoss(loss_fn, [data_qhbm.e_inference], model_vars, exact_inferences=[model.e_inference])
  multiset = _multiset(data_qhbm.e_inference, samples)
  want_loss = step.eager([multiset]).clone()
  want = _grads(model_vars)
  got = step([multiset])
  torch.cuda.synchronize()
  assert torch.equal(got, want_loss) and all(torch.equal(v.grad, w) for v, w in zip(model_vars, want))
  assert any(float(w.abs().max()) > 0 for w in want)


def test_capture_refuses_host_resident_variables():
  qubits = ir.GridQubit.rect(1, 3)
  circuit = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "cpu"))          # parameters on the host
  energy = models.BernoulliEnergy([0, 1, 2]).to("cuda")
  qhbm = inference.QHBM(inference.BernoulliEnergyInference(energy, 8, initial_seed=1), inference.AnalyticQuantumInference(circuit))
  with pytest.raises(E.EngineError, match="ONE CUDA device"):
    inference.CapturedLoss(lambda: inference.vqt(qhbm, [_tfim(qubits)], 1.0), [qhbm.e_inference],
                           list(energy.parameters()) + circuit.trainable_variables)
