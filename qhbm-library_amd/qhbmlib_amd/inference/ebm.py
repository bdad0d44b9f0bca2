"""Inference on energy-based models (reference: qhbmlib/inference/ebm.py).

Classical side of the loop: it PRODUCES the bitstring batch the hot path consumes
and CALLS it (ebm.py:278).  Only what closes the VQT / QMHL loop is mirrored:
the base class contract, the score-function gradient of `expectation`
(ebm.py:262-329), and the analytic / Bernoulli samplers.  Samplers match the
reference distributionally (its tests are statistical), not
bit-for-bit: TFP's stateless PRNG is not reproduced.
"""
import abc
import contextlib
import itertools
from typing import Union

import torch

from qhbmlib_amd import parallel
from qhbmlib_amd import utils
from qhbmlib_amd.models import energy


def fresh_seed() -> int:
  """The seed of a sampler built with `initial_seed=None` (the reference draws one with
  `tfp.random.sanitize_seed`, ebm.py:74-79): taken from torch's global generator, so that
  `torch.manual_seed` makes a run repeatable.  LOCAL: no communication happens at construction time.  Ranks
  that shard one expectation must draw the same samples; their samplers are given rank 0's seed by
  `EnergyInferenceBase.agree_seed` -- called once, lazily, by `QHBM.agree_seeds()` inside the first sharded
  call, or explicitly through `parallel.agree_seeds(...)`.  Ranks that never shard keep independent seeds."""
  return int(torch.randint(0, 2**31 - 1, (), dtype=torch.int64).item())


class EnergyInferenceBase(torch.nn.Module, abc.ABC):
  """Interface for inference on BitstringEnergy objects (ebm.py:48-230)."""

  def __init__(self, input_energy: energy.BitstringEnergy,
               initial_seed: Union[None, int] = None, name: Union[None, str] = None):
    super().__init__()
    self.name = name or type(self).__name__
    self._energy = input_energy
    self._energy.build([None, self._energy.num_bits])
    self._tracked_variables = list(input_energy.parameters())
    self._checkpoint = [v.detach().clone() for v in self._tracked_variables]
    self._update_seed = initial_seed is None
    self._generator = torch.Generator()
    self._seed = fresh_seed() if initial_seed is None else int(initial_seed)
    self._first_inference = True
    self._fixed_multiset = None
    self._device_only = 0

  @contextlib.contextmanager
  def device_only(self):
    """Inside the block `_preface_inference` is skipped: no variable is read back to the host to see whether it changed
    (ebm.py:125-134 compares with a checkpoint), no seed is advanced, nothing is made ready.  For inferences a recorded
    step (`inference.CapturedLoss`) only asks for EXACT quantities computed from the live variables -- `log_partition` /
    `entropy` of the analytic and Bernoulli inferences -- and never for samples."""
    self._device_only += 1
    try:
      yield self
    finally:
      self._device_only -= 1

  @contextlib.contextmanager
  def fixed_samples(self, bitstrings: torch.Tensor, counts: torch.Tensor):
    """Inside the block, sample averages (`expectation`, the gradient of the default `log_partition`) run over the GIVEN
    multiset -- rows `bitstrings[i]` with multiplicities `counts[i]` (zero-count rows are allowed and contribute nothing)
    -- instead of drawing `num_expectation_samples` samples and deduplicating them (ebm.py:271-273).  Two uses: timing a
    step with the sampler excluded (`bench.py --through-mirror`, BASELINE.md section 3 "VQT step time"), and the
    fixed-shape, synchronisation-free step that `inference.CapturedLoss` records into a hipGraph: no sample is drawn, no
    variable is read back to the host (`_preface_inference` is skipped), nothing is sorted."""
    previous = self._fixed_multiset
    self._fixed_multiset = (bitstrings, counts)
    try:
      yield self
    finally:
      self._fixed_multiset = previous

  def _unique_samples(self, num_samples):
    """(unique bitstrings, counts) of `num_samples` model samples (ebm.py:271-273, 402-403), or the fixed multiset."""
    if self._fixed_multiset is not None:
      bitstrings, counts = self._fixed_multiset
      utils.mark_rows_unique(bitstrings)   # (the quantum side must not sort them again; duplicates would only cost time)
      return bitstrings, counts
    with torch.no_grad():
      samples = self.sample(num_samples)
    bitstrings, _, counts = utils.unique_bitstrings_with_counts(samples)
    return bitstrings, counts

  @property
  def energy(self):
    return self._energy

  @property
  def seed(self):
    return self._seed

  @seed.setter
  def seed(self, initial_seed):
    self._update_seed = initial_seed is None
    if initial_seed is not None:
      self._seed = int(initial_seed)

  def agree_seed(self, group=None):
    """COLLECTIVE over `group` (every rank takes part in the broadcast): a sampler built with `initial_seed=None`
    takes rank 0's current seed (parallel.agreed_seed), on the device of the energy's variables when the group's
    backend moves device memory.  An EXPLICIT seed is the caller's: it is kept, and ranks whose explicit seeds
    differ fail loudly in the consistency check of the sharded call instead of being re-seeded silently.
    Returns True when the seed was replaced."""
    agreed = parallel.agreed_seed(self._seed, group, device=_device_of(self._energy))
    if not self._update_seed or agreed == self._seed:
      return False
    self._seed = agreed
    return True

  @property
  def variables_updated(self):
    """True if tracked variables differ from the checkpointed values (ebm.py:125-134): compared by VALUE, as the
    reference does (a version counter would miss writes through `.data`), where the variable lives -- one boolean comes
    back to the host per variable, not the variable."""
    return any(not torch.equal(v.detach(), c.to(v.device)) for v, c in zip(self._tracked_variables, self._checkpoint))

  def _checkpoint_variables(self):
    self._checkpoint = [v.detach().clone() for v in self._tracked_variables]

  def _preface_inference(self):
    """ebm.py:142-162."""
    if self._fixed_multiset is not None or self._device_only:
      return  # (`fixed_samples`: nothing is drawn, so nothing has to be made ready -- and no host read of the variables)
    if self._first_inference:
      self._checkpoint_variables()
      self._ready_inference()
      self._first_inference = False
    if self._update_seed:
      self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) % (2**63)
    if self.variables_updated:
      self._checkpoint_variables()
      self._ready_inference()

  def _rng(self):
    return self._generator.manual_seed(self._seed % (2**63))

  @abc.abstractmethod
  def _ready_inference(self):
    """Recomputes cached quantities after the energy's variables changed."""

  def forward(self, inputs=None):
    self._preface_inference()
    return self._call(inputs)

  def entropy(self):
    self._preface_inference()
    return self._entropy()

  def expectation(self, function):
    self._preface_inference()
    return self._expectation(function)

  def log_partition(self):
    self._preface_inference()
    return self._log_partition()

  def sample(self, num_samples: int):
    self._preface_inference()
    return self._sample(num_samples)

  @abc.abstractmethod
  def _call(self, inputs):
    raise NotImplementedError()

  @abc.abstractmethod
  def _entropy(self):
    raise NotImplementedError()

  @abc.abstractmethod
  def _expectation(self, function):
    raise NotImplementedError()

  @abc.abstractmethod
  def _log_partition(self):
    raise NotImplementedError()

  @abc.abstractmethod
  def _sample(self, num_samples: int):
    raise NotImplementedError()


class EnergyInference(EnergyInferenceBase):
  """Default implementations by sample averaging (ebm.py:233-415)."""

  def __init__(self, input_energy: energy.BitstringEnergy, num_expectation_samples: int,
               initial_seed: Union[None, int] = None, name: Union[None, str] = None):
    super().__init__(input_energy, initial_seed, name)
    self.num_expectation_samples = num_expectation_samples

  def _entropy(self):
    return self.expectation(self.energy) + self.log_partition()

  def _expectation(self, function):
    """Sample average with the gradient of ebm.py:262-329 (eq. A5 of the QHBM paper):
        d<f> = <df> + <f><dE> - <f dE>
    realised with a zero-valued surrogate whose derivative is the score-function term."""
    bitstrings, counts = self._unique_samples(self.num_expectation_samples)
    values = function(bitstrings)
    single = torch.is_tensor(values)
    flat = [values] if single else list(values)
    energies = self.energy(bitstrings.to(next(iter(self.energy.parameters()), torch.zeros(())).device))
    out = []
    for v in flat:
      avg = utils.weighted_average(counts, v)
      w = (counts.to(torch.float32) / counts.sum()).to(v.device)
      centered = (v.detach() - avg.detach()).reshape(v.shape[0], -1)
      surrogate = -torch.tensordot(w * energies.to(v.device), centered, dims=([0], [0])).reshape(avg.shape)
      out.append(avg + (surrogate - surrogate.detach()))
    return out[0] if single else type(values)(out)

  def _log_partition(self):
    """Default estimator (ebm.py:331-415): the VALUE is the Monte-Carlo estimate over uniform samples
    (`_log_partition_forward_pass`, ebm.py:345-394), the GRADIENT is the reference's own estimator, eq. C2 of the
    QHBM paper -- minus the average of dE under MODEL samples (`_log_partition_grad_generator`, ebm.py:396-415) --
    not the derivative of the forward estimate (which would be self-normalised importance sampling from the uniform
    distribution: useless for a peaked distribution over many bits)."""
    variables = [v for v in self.energy.parameters() if v.requires_grad]
    return _LogPartition.apply(self, *variables)

  def _log_partition_forward_pass(self):
    """log Z ~ n log 2 - log N_s + logsumexp(-E(x_i)), x_i uniform (ebm.py:345-394)."""
    n = self.energy.num_bits
    n_s = self.num_expectation_samples
    samples = torch.randint(0, 2, (n_s, n), generator=self._rng(), dtype=torch.int8)
    energies = self.energy(samples.to(_device_of(self.energy)))
    return (n * torch.log(torch.tensor(2.0)) - torch.log(torch.tensor(float(n_s))) +
            torch.logsumexp(-1.0 * energies, 0))


def _as_rows(x):
  """A 1-D tensor of 2^n entries as [rows, columns] with columns <= 2^10: reductions along the columns, then over the
  rows, are each the work of ONE thread block per output -- no multi-block reduction with global semaphores, whose
  zero-fill a hipGraph capture records as a memset node (see `inference.CapturedLoss`).  The values are the same sums in
  another association order (a gradient flows through both stages)."""
  n = x.numel()
  cols = 1
  while cols < 1024 and n % (cols * 2) == 0:
    cols *= 2
  return x.reshape(n // cols, cols)


def _logsumexp_rows(x):
  return torch.logsumexp(torch.logsumexp(_as_rows(x), 1), 0)


def _sum_rows(x):
  return _as_rows(x).sum(1).sum(0)


def _device_of(module):
  return next(iter(module.parameters()), torch.zeros(())).device


class _LogPartition(torch.autograd.Function):
  """tf.custom_gradient of ebm.py:331-343 as a torch.autograd.Function over the energy's variables."""

  @staticmethod
  def forward(ctx, inference, *variables):   # pylint: disable=arguments-differ
    ctx.inference = inference
    ctx.variables = variables
    with torch.no_grad():
      return inference._log_partition_forward_pass()   # pylint: disable=protected-access

  @staticmethod
  def backward(ctx, upstream):   # pylint: disable=arguments-differ
    inf, variables = ctx.inference, ctx.variables
    if not variables:
      return (None,)
    unique_samples, counts = inf._unique_samples(inf.num_expectation_samples)   # ebm.py:402-403 (model samples)
    with torch.enable_grad():
      unique_energies = inf.energy(unique_samples.to(_device_of(inf.energy)))
      weights = (counts.to(torch.float32) / counts.sum()).to(unique_energies.device)
      average = torch.sum(weights * unique_energies)               # weighted average of the Jacobian rows, ebm.py:411-412
      grads = torch.autograd.grad(average, variables, allow_unused=True)
    return (None,) + tuple(torch.zeros_like(v) if g is None else -1.0 * upstream.to(g.device) * g
                           for v, g in zip(variables, grads))


class AnalyticEnergyInference(EnergyInference):
  """Explicit categorical distribution over all bitstrings (ebm.py:418-492).

  The 2^n bitstrings, their energies and the categorical live on the device of the energy's
  variables: with the energy on the GPU, `_ready_inference` over 2^20 bitstrings of a KOBE is one
  `qhbm_parity_energy` launch (the reference re-evaluates a Python loop over the parity terms,
  `energy_utils.py:106-110`, after every variable update)."""

  def __init__(self, input_energy: energy.BitstringEnergy, num_expectation_samples: int,
               initial_seed: Union[None, int] = None, name: Union[None, str] = None):
    super().__init__(input_energy, num_expectation_samples, initial_seed, name)
    n = input_energy.num_bits
    # rows in itertools.product([0, 1], repeat=n) order (ebm.py:445-447)
    index = torch.arange(2**n, dtype=torch.int64).unsqueeze(1)
    shifts = torch.arange(n - 1, -1, -1, dtype=torch.int64).unsqueeze(0)
    self._all_bitstrings = ((index >> shifts) & 1).to(torch.int8)
    self._logits = None
    self._device_generator = None

  def _device(self):
    return next(iter(self.energy.parameters()), torch.zeros(())).device

  @property
  def all_bitstrings(self):
    dev = self._device()
    if self._all_bitstrings.device != dev:
      self._all_bitstrings = self._all_bitstrings.to(dev)
    return self._all_bitstrings

  @property
  def all_energies(self):
    return self.energy(self.all_bitstrings)

  @property
  def distribution(self):
    """The categorical over all bitstrings (ebm.py:462-465)."""
    return torch.distributions.Categorical(logits=self._logits)

  def _ready_inference(self):
    with torch.no_grad():
      self._logits = -self.all_energies.detach()

  def _call(self, inputs):
    if inputs is None:
      return self.distribution
    return self.sample(inputs)

  def _entropy(self):
    logits = -self.all_energies
    logp = logits - _logsumexp_rows(logits)
    plogp = torch.exp(logp) * logp
    return -_sum_rows(plogp)

  def _log_partition(self):
    return _logsumexp_rows(-self.all_energies)

  def _sample(self, num_samples: int):
    probs = torch.softmax(self._logits.to(torch.float64), 0)
    if probs.is_cuda:
      if self._device_generator is None or self._device_generator.device != probs.device:
        self._device_generator = torch.Generator(device=probs.device)
      gen = self._device_generator.manual_seed(self._seed % (2**63))
    else:
      gen = self._rng()
    if probs.numel() <= 2**24:
      idx = torch.multinomial(probs, num_samples, replacement=True, generator=gen)
    else:  # torch.multinomial stops at 2^24 categories: inverse-CDF sampling
      u = torch.rand(num_samples, dtype=torch.float64, device=probs.device, generator=gen)
      idx = torch.searchsorted(torch.cumsum(probs, 0), u).clamp_(max=probs.numel() - 1)
    return self.all_bitstrings[idx]


class BernoulliEnergyInference(EnergyInference):
  """Inference for a Bernoulli defined by spin energies (ebm.py:495-561)."""

  def __init__(self, input_energy: energy.BernoulliEnergy, num_expectation_samples: int,
               initial_seed: Union[None, int] = None, name: Union[None, str] = None):
    super().__init__(input_energy, num_expectation_samples, initial_seed, name)
    self._logits = None

  def _ready_inference(self):
    with torch.no_grad():
      self._logits = self.energy.logits.detach().cpu().clone()

  def _call(self, inputs):
    if inputs is None:
      return torch.distributions.Bernoulli(logits=self._logits)
    return self.sample(inputs)

  def _entropy(self):
    """Sum of the per-spin entropies (exact; ebm.py:537-544)."""
    logits = self.energy.logits
    p = torch.sigmoid(logits)
    return torch.sum(torch.nn.functional.softplus(logits) - p * logits)

  def _log_partition(self):
    """sum_i log(e^theta_i + e^-theta_i) (exact; ebm.py:546-557)."""
    thetas = 0.5 * self.energy.logits
    return torch.sum(torch.log(torch.exp(thetas) + torch.exp(-thetas)))

  def _sample(self, num_samples: int):
    p = torch.sigmoid(self._logits).expand(num_samples, -1)
    return torch.bernoulli(p, generator=self._rng()).to(torch.int8)


class GibbsWithGradientsKernel:
  """The Gibbs-With-Gradients update rule, Algorithm 1 of arXiv:2102.04509v2 (ebm.py:564-702):
  a Metropolis-Hastings chain over bitstrings whose index proposal q(i | x) is the softmax of the
  first-order estimate of the energy change of flipping bit i, d(x) ~ (2x - 1) * dE(x)/dx."""

  def __init__(self, input_energy: energy.BitstringEnergy, generator: Union[None, torch.Generator] = None):
    self._energy = input_energy
    self._num_bits = input_energy.num_bits
    self._parameters = dict(input_energy=input_energy)
    self._generator = generator

  @property
  def is_calibrated(self):
    """True: the chain converges to the distribution of the energy (ebm.py:687-690)."""
    return True

  def bootstrap_results(self, init_state):
    del init_state
    return []

  def _device(self):
    """Where the energy's variables live: the chain state stays on the host (its random stream is a
    host generator), energy evaluations run on the energy's device."""
    first = next(iter(self._energy.parameters()), None)
    return torch.device("cpu") if first is None else first.device

  def _get_index_proposal_probs(self, x):
    """Equation 6 of the paper with the Taylor estimate of equation 3 (ebm.py:618-650)."""
    x_float = torch.as_tensor(x).to(device=self._device(), dtype=torch.float32).detach().clone().requires_grad_(True)
    with torch.enable_grad():
      current_energy = self._energy(x_float.unsqueeze(0)).squeeze()
      (e_grad,) = torch.autograd.grad(current_energy, x_float)
    f_grad = -1.0 * e_grad  # f(x) = -E(x)
    approx_energy_diff = -(2.0 * x_float.detach() - 1.0) * f_grad
    return torch.softmax(approx_energy_diff / 2.0, 0)

  def one_step(self, current_state, previous_kernel_results):
    """One Metropolis-Hastings step (ebm.py:652-685): returns (next_state, [])."""
    del previous_kernel_results
    current_state = torch.as_tensor(current_state).to(torch.int8)
    with torch.no_grad():
      current_state = current_state.cpu()
      probs = self._get_index_proposal_probs(current_state).cpu()
      i = int(torch.multinomial(probs, 1, generator=self._generator))
      x_prime = current_state.clone()
      x_prime[i] = 1 - x_prime[i]
      probs_prime = self._get_index_proposal_probs(x_prime).cpu()
      q_ratio = probs_prime[i] / probs[i]
      energies = self._energy(torch.stack([x_prime, current_state]).to(self._device())).cpu()
      accept_prob = torch.clamp(torch.exp(-energies[0] + energies[1]) * q_ratio, max=1.0)
      roll = torch.rand((), generator=self._generator)
      next_state = x_prime if bool(roll <= accept_prob.cpu()) else current_state
    return next_state, []


class GibbsWithGradientsInference(EnergyInference):
  """Inference with a Gibbs-With-Gradients Markov chain (ebm.py:705-760): the chain state
  persists between calls; `num_burnin_samples` steps are discarded whenever the energy's
  variables have changed."""

  def __init__(self, input_energy: energy.BitstringEnergy, num_expectation_samples: int,
               num_burnin_samples: int, name: Union[None, str] = None,
               initial_seed: Union[None, int] = None):
    super().__init__(input_energy, num_expectation_samples, initial_seed, name)
    # the chain owns its random stream (the base class re-seeds the shared generator per call)
    self._chain_generator = torch.Generator().manual_seed(self._seed % (2**63))
    self._kernel = GibbsWithGradientsKernel(input_energy, self._chain_generator)
    self._chain_state = torch.bernoulli(torch.full((self.energy.num_bits,), 0.5),
                                        generator=self._chain_generator).to(torch.int8)
    self.num_burnin_samples = num_burnin_samples

  def agree_seed(self, group=None):
    """... and the chain restarts from the agreed seed (its generator and initial state were drawn from the old one)."""
    if not super().agree_seed(group):
      return False
    self._chain_generator.manual_seed(self._seed % (2**63))
    self._chain_state = torch.bernoulli(torch.full((self.energy.num_bits,), 0.5),
                                        generator=self._chain_generator).to(torch.int8)
    self._first_inference = True   # burn in again
    return True

  def _ready_inference(self):
    state = self._chain_state
    for _ in range(int(self.num_burnin_samples)):
      state, _ = self._kernel.one_step(state, [])
    self._chain_state = state

  def _call(self, inputs):
    return self.sample(inputs)

  def _sample(self, num_samples: int):
    out = torch.empty((int(num_samples), self.energy.num_bits), dtype=torch.int8)
    state = self._chain_state
    for i in range(int(num_samples)):
      state, _ = self._kernel.one_step(state, [])
      out[i] = state
    self._chain_state = state
    return out
