"""Energy functions over bitstrings (reference: qhbmlib/models/energy.py)."""
import abc
from typing import List, Union

import numpy as np
import torch

from qhbmlib_amd import ir
from qhbmlib_amd.models import energy_utils


class BitstringEnergy(torch.nn.Module):
  """E(x): unnormalised negative log-probability of a bitstring (energy.py:26-87)."""

  def __init__(self, bits: List[int], energy_layers: List[torch.nn.Module],
               name: Union[None, str] = None):
    super().__init__()
    self.name = name or type(self).__name__
    self._bits = energy_utils.check_bits(bits)
    self._energy_layers = torch.nn.ModuleList(energy_layers)

  @property
  def num_bits(self):
    return len(self.bits)

  @property
  def bits(self):
    return self._bits

  @property
  def energy_layers(self):
    return list(self._energy_layers)

  def build(self, input_shape):
    x = list(input_shape)
    for layer in self._energy_layers:
      if hasattr(layer, "compute_output_shape"):
        x = layer.compute_output_shape(x)

  @property
  def trainable_variables(self):
    return [p for p in self.parameters() if p.requires_grad]

  def forward(self, inputs):
    x = inputs
    for layer in self._energy_layers:
      x = layer(x)
    return x


class PauliMixin(abc.ABC):
  """Adds a Pauli-Z representation to a BitstringEnergy (energy.py:90-120)."""

  @property
  @abc.abstractmethod
  def post_process(self):
    raise NotImplementedError()

  @abc.abstractmethod
  def operator_shards(self, qubits):
    raise NotImplementedError()

  def operator_expectation(self, expectation_shards: torch.Tensor):
    """Average energy from the shard expectations (energy.py:115-120)."""
    x = expectation_shards
    for layer in self.post_process:
      x = layer(x)
    return x

  # ---- GPU path of the spin-parity energies ----------------------------------
  def _parity_index_sets(self):
    """Column index sets whose spin products the energy weights, in kernel order."""
    raise NotImplementedError()

  def _parity_masks(self, device):
    cache = self.__dict__.setdefault("_mask_cache", {})
    key = str(device)
    if key not in cache:
      masks = [sum(1 << int(c) for c in ix) for ix in self._parity_index_sets()]
      # uint64 on the kernel side; column 63 sets the sign bit of the int64 torch stores it in
      as_u64 = np.asarray(masks, dtype=np.uint64)
      cache[key] = torch.from_numpy(as_u64.view(np.int64).copy()).to(device)
    return cache[key]

  def _gpu_energy(self, inputs):
    """One HIP kernel instead of SpinsFromBitstrings -> Parity -> VariableDot when the
    bitstrings are integer CUDA tensors with at most 64 columns (qhbm_parity_energy)."""
    from qhbmlib_amd import _engine  # pylint: disable=import-outside-toplevel
    flat = inputs.reshape(-1, inputs.shape[-1])
    kernel = self.post_process[0].kernel
    e = _engine.parity_energy(kernel, flat, self._parity_masks(flat.device))
    return e.reshape(inputs.shape[:-1])

  def _use_gpu_energy(self, inputs):
    return (torch.is_tensor(inputs) and inputs.is_cuda and not inputs.is_floating_point() and
            inputs.shape[-1] <= 64 and self.post_process[0].kernel is not None)


class BernoulliEnergy(BitstringEnergy, PauliMixin):
  """Tensor product of coin flips, E(b) = sum_i theta_i (1 - 2 b_i) (energy.py:123-167)."""

  def __init__(self, bits: List[int], initializer=None, name: Union[None, str] = None):
    pre_process = [energy_utils.SpinsFromBitstrings()]
    post_process = [energy_utils.VariableDot(initializer=initializer)]
    super().__init__(bits, pre_process + post_process, name)
    self._post_process = post_process
    self.build([None, self.num_bits])

  @property
  def logits(self):
    """logit = log p/(1-p) = 2 theta (energy.py:148-158)."""
    return 2 * self.post_process[0].kernel

  @property
  def post_process(self):
    return self._post_process

  def operator_shards(self, qubits):
    """energy.py:165-167."""
    return [ir.PauliSum.from_pauli_strings(ir.PZ(q)) for q in qubits]

  def _parity_index_sets(self):
    return [(i,) for i in range(self.num_bits)]

  def forward(self, inputs):
    if self._use_gpu_energy(inputs):
      return self._gpu_energy(inputs)
    return super().forward(inputs)


class KOBE(BitstringEnergy, PauliMixin):
  """Kth Order Binary Energy function (energy.py:170-209)."""

  def __init__(self, bits: List[int], order: int, initializer=None,
               name: Union[None, str] = None):
    parity_layer = energy_utils.Parity(bits, order)
    self._num_terms = parity_layer.num_terms
    self._indices = parity_layer.indices
    pre_process = [energy_utils.SpinsFromBitstrings(), parity_layer]
    post_process = [energy_utils.VariableDot(initializer=initializer)]
    super().__init__(bits, pre_process + post_process, name)
    self._post_process = post_process
    self.build([None, self.num_bits])

  @property
  def post_process(self):
    return self._post_process

  def operator_shards(self, qubits):
    """energy.py:200-209."""
    ops = []
    for i in range(self._num_terms):
      string = ir.PauliString(*[ir.PZ(qubits[loc]) for loc in self._indices[i]])
      ops.append(ir.PauliSum.from_pauli_strings(string))
    return ops

  def _parity_index_sets(self):
    return self._indices

  def forward(self, inputs):
    if self._use_gpu_energy(inputs):
      return self._gpu_energy(inputs)
    return super().forward(inputs)
