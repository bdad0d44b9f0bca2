"""Utilities for the energy module (reference: qhbmlib/models/energy_utils.py)."""
import itertools
from typing import List

import torch


def check_bits(bits: List[int]) -> List[int]:
  """energy_utils.py:23-27."""
  if len(set(bits)) != len(bits):
    raise ValueError("All entries of `bits` must be unique.")
  return bits


def check_order(order: int) -> int:
  """energy_utils.py:30-36."""
  if not isinstance(order, int):
    raise TypeError("`order` must be an integer.")
  if order <= 0:
    raise ValueError("`order` must be greater than zero.")
  return order


class SpinsFromBitstrings(torch.nn.Module):
  """|0> -> +1, |1> -> -1 (energy_utils.py:39-52)."""

  def forward(self, inputs):
    return (1 - 2 * inputs).to(torch.float32)


class VariableDot(torch.nn.Module):
  """Dot product with a same-sized trainable kernel (energy_utils.py:55-81)."""

  def __init__(self, initializer=None):
    super().__init__()
    self._initializer = initializer
    self.kernel = None

  def build(self, input_shape):
    if self.kernel is None:
      n = int(input_shape[-1])
      init = self._initializer or (lambda shape: torch.empty(shape).uniform_(-0.05, 0.05))
      self.kernel = torch.nn.Parameter(torch.as_tensor(init([n]), dtype=torch.float32).clone())

  def compute_output_shape(self, input_shape):
    self.build(input_shape)
    return list(input_shape[:-1])

  def forward(self, inputs):
    self.build(inputs.shape)
    return torch.sum(inputs * self.kernel.to(inputs.device), -1)


class Parity(torch.nn.Module):
  """Parities of all bit groups of size 1..order, in itertools.combinations order
  (energy_utils.py:84-110)."""

  def __init__(self, bits: List[int], order: int):
    super().__init__()
    bits = check_bits(bits)
    order = check_order(order)
    indices_list = []
    for i in range(1, order + 1):
      indices_list.extend(list(itertools.combinations(range(len(bits)), i)))
    self.indices = indices_list
    self.num_terms = len(indices_list)

  def compute_output_shape(self, input_shape):
    return list(input_shape[:-1]) + [self.num_terms]

  def forward(self, inputs):
    cols = [torch.prod(inputs[..., list(ix)], dim=-1) for ix in self.indices]
    return torch.stack(cols, dim=-1)
