"""The DEFAULT estimators of EnergyInference (entropy, expectation, log-partition and their gradients), ported from
the reference's EnergyInferenceTest (/root/reference/tests/inference/ebm_test.py:30-175): a subclass that only
supplies a Bernoulli sampler, 5 bits, 10^6 samples, against exact values and finite differences at the reference's
tolerances (rtol 2e-2, atol 2e-3).  Device-agnostic torch code: runs on the CPU."""
import itertools

import numpy as np
import torch

from qhbmlib_amd import inference, models, utils

CLOSE_RTOL, CLOSE_ATOL, NOT_ZERO_ATOL = 2e-2, 2e-3, 4e-3
NUM_SAMPLES = int(1e6)
NUM_BITS = 5


class EnergyInferenceBernoulliSampler(inference.EnergyInference):
  """EnergyInference whose sampler is just a Bernoulli (ebm_test.py:33-55)."""

  def __init__(self, energy, num_expectation_samples, initial_seed):
    super().__init__(energy, num_expectation_samples, initial_seed)
    self._logits = energy.logits.detach().clone()

  def _ready_inference(self):
    self._logits = self.energy.logits.detach().clone()

  def _call(self, inputs):
    return self.sample(inputs)

  def _sample(self, num_samples):
    p = torch.sigmoid(self._logits).expand(int(num_samples), -1)
    return torch.bernoulli(p, generator=self._rng()).to(torch.int8)


def _setup(seed=4):
  torch.manual_seed(seed)
  energy = models.BernoulliEnergy(list(range(NUM_BITS)))
  energy.build([None, NUM_BITS])
  with torch.no_grad():   # RandomUniform(-0.05, 0.05) is Keras' default range; the reference seeds it (ebm_test.py:70-73)
    for v in energy.parameters():
      v.uniform_(-0.05, 0.05)
  ebm = EnergyInferenceBernoulliSampler(energy, NUM_SAMPLES, 34)
  spins = models.SpinsFromBitstrings()
  parity = models.Parity(list(range(NUM_BITS)), 2)
  return energy, ebm, (lambda bitstrings: parity(spins(bitstrings)))


def _approximate_gradient(f, variables, delta=1e-1):
  """Five-point stencil, tests/test_util.py:186-243 of the reference."""
  grads = []
  for v in variables:
    g = torch.zeros_like(v)
    flat = v.data.view(-1)
    for i in range(flat.numel()):
      old = float(flat[i])
      vals = []
      for k in (2, 1, -1, -2):
        flat[i] = old + k * delta
        with torch.no_grad():
          vals.append(f())
      flat[i] = old
      g.view(-1)[i] = (-vals[0] + 8 * vals[1] - 8 * vals[2] + vals[3]) / (12 * delta)
    grads.append(g)
  return grads


def test_entropy_value_and_gradient():
  """ebm_test.py:87-108."""
  energy, ebm, _ = _setup()

  def manual_entropy():
    return torch.sum(torch.distributions.Bernoulli(logits=energy.logits).entropy())

  np.testing.assert_allclose(float(ebm.entropy()), float(manual_entropy()), rtol=CLOSE_RTOL)
  expected = _approximate_gradient(manual_entropy, list(energy.parameters()))
  value = ebm.entropy()
  actual = torch.autograd.grad(value, list(energy.parameters()))
  for a, e in zip(actual, expected):
    np.testing.assert_allclose(a.numpy(), e.numpy(), rtol=CLOSE_RTOL, atol=CLOSE_ATOL)


def test_expectation_value_and_gradient():
  """ebm_test.py:110-136: exact expectation over all bitstrings as the expected side (the reference draws the same
  seeded samples twice; TFP's stateless stream is not reproduced here)."""
  energy, ebm, f = _setup()
  all_bits = torch.tensor(list(itertools.product([0, 1], repeat=NUM_BITS)), dtype=torch.int8)

  def manual_expectation():
    logp = -energy(all_bits)
    p = torch.softmax(logp, 0)
    return torch.tensordot(p, f(all_bits), dims=([0], [0]))

  np.testing.assert_allclose(ebm.expectation(f).detach().numpy(), manual_expectation().detach().numpy(),
                             rtol=CLOSE_RTOL, atol=CLOSE_ATOL)
  # gradient of the sum of the outputs (tape.gradient of a vector value sums it)
  expected = _approximate_gradient(lambda: manual_expectation().sum(), list(energy.parameters()))
  value = ebm.expectation(f).sum()
  actual = torch.autograd.grad(value, list(energy.parameters()))
  for a, e in zip(actual, expected):
    np.testing.assert_allclose(a.numpy(), e.numpy(), rtol=CLOSE_RTOL, atol=5 * CLOSE_ATOL)


def test_log_partition_value_and_gradient():
  """ebm_test.py:138-175: value from the uniform-sample estimate, gradient = -<dE> under model samples."""
  energy, ebm, _ = _setup()
  all_bits = torch.tensor(list(itertools.product([0, 1], repeat=NUM_BITS)), dtype=torch.int8)

  def manual_log_partition():
    return torch.logsumexp(-1.0 * energy(all_bits), 0)

  np.testing.assert_allclose(float(ebm.log_partition()), float(manual_log_partition()), rtol=CLOSE_RTOL)
  # a distribution with a gradient that is not ~0 (the reference asserts this of its initialiser, ebm_test.py:164-167)
  with torch.no_grad():
    for v in energy.parameters():
      v.copy_(torch.linspace(-0.6, 0.8, v.numel()).reshape(v.shape))
  expected = _approximate_gradient(manual_log_partition, list(energy.parameters()))
  assert max(float(e.abs().max()) for e in expected) > NOT_ZERO_ATOL
  value = ebm.log_partition()
  actual = torch.autograd.grad(value, list(energy.parameters()))
  for a, e in zip(actual, expected):
    np.testing.assert_allclose(a.numpy(), e.numpy(), atol=CLOSE_ATOL)


def test_log_partition_gradient_uses_model_samples_not_the_uniform_reweighting():
  """A peaked 12-bit distribution: differentiating the uniform-sample estimate (round 3's mirror) is far off with
  2000 samples, the reference's estimator -<dE>_model is not."""
  torch.manual_seed(0)
  n = 12
  energy = models.BernoulliEnergy(list(range(n)))
  energy.build([None, n])
  with torch.no_grad():
    for v in energy.parameters():
      v.fill_(2.5)
  ebm = EnergyInferenceBernoulliSampler(energy, 2000, 7)
  thetas = next(energy.parameters())
  exact = torch.tanh(thetas.detach())          # d/dtheta sum log(2 cosh theta)
  got = torch.autograd.grad(ebm.log_partition(), [thetas])[0]
  np.testing.assert_allclose(got.numpy().ravel(), exact.numpy().ravel(), atol=3e-2)


def test_gibbs_with_gradients_log_partition_gradient_matches_analytic():
  """GibbsWithGradientsInference (the one class that uses the default) against the analytic
  gradient at 5 bits."""
  torch.manual_seed(3)
  n = 5
  energy = models.KOBE(list(range(n)), 2)
  energy.build([None, n])
  with torch.no_grad():
    for v in energy.parameters():
      v.uniform_(-0.4, 0.4)
  exact_ebm = inference.AnalyticEnergyInference(energy, 10, initial_seed=1)
  want = torch.autograd.grad(exact_ebm.log_partition(), list(energy.parameters()))
  gwg = inference.GibbsWithGradientsInference(energy, 20000, 200, initial_seed=5)
  got = torch.autograd.grad(gwg.log_partition(), list(energy.parameters()))
  for a, e in zip(got, want):
    np.testing.assert_allclose(a.numpy(), e.numpy(), atol=4e-2)


def test_fixed_samples_average_over_a_given_multiset_and_ignore_zero_count_rows():
  """`EnergyInferenceBase.fixed_samples(bitstrings, counts)`: the sample average of ebm.py:262-329 and its score-function
  gradient over a GIVEN multiset (what `inference.CapturedLoss` records and `bench.py --through-mirror` times); rows with
  count 0 -- the padding of a fixed-capacity buffer -- change nothing; no sample is drawn, no seed advances."""
  import torch
  from qhbmlib_amd import inference, models, utils
  torch.manual_seed(3)
  energy = models.KOBE(list(range(5)), 2)
  with torch.no_grad():
    energy.post_process[0].kernel.uniform_(-0.7, 0.7)
  e_inf = inference.AnalyticEnergyInference(energy, 64, initial_seed=9)
  drawn = e_inf.sample(64)
  rows, _, counts = utils.unique_bitstrings_with_counts(drawn)
  seed_before = e_inf.seed
  f = lambda b: (b.to(torch.float32) * torch.arange(1.0, 6.0)).sum(1)
  theta = energy.post_process[0].kernel

  def value_and_grad(r, c):
    with e_inf.fixed_samples(r, c):
      out = e_inf.expectation(f)
    (g,) = torch.autograd.grad(out, [theta])
    return out.detach(), g

  got, grad = value_and_grad(rows, counts)
  w = counts.to(torch.float32) / counts.sum()
  np.testing.assert_allclose(float(got), float((w * f(rows)).sum()), rtol=1e-6)
  # d<f> = <f><dE> - <f dE> with dE/dtheta_k = the parity features of the rows
  from oracle import qhbm_oracle as O
  feats = torch.from_numpy(O.parities(rows.numpy(), O.parity_indices(5, 2)).astype(np.float32))   # [U, 15] = dE/dtheta
  want = (w * f(rows)).sum() * (w[:, None] * feats).sum(0) - (w[:, None] * f(rows)[:, None] * feats).sum(0)
  np.testing.assert_allclose(grad.numpy(), want.numpy(), atol=1e-5)
  padded_rows = torch.cat([rows, rows[:1].expand(7, -1)])
  padded_counts = torch.cat([counts, torch.zeros(7, dtype=counts.dtype)])
  got_p, grad_p = value_and_grad(padded_rows, padded_counts)
  np.testing.assert_allclose(float(got_p), float(got), rtol=1e-6)
  np.testing.assert_allclose(grad_p.numpy(), grad.numpy(), atol=1e-6)
  assert e_inf.seed == seed_before and e_inf._fixed_multiset is None
