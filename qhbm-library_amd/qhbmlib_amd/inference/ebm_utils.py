"""Utilities for metrics on BitstringEnergy (reference: qhbmlib/inference/ebm_utils.py)."""
import itertools

import torch

from qhbmlib_amd.models import energy


def probabilities(input_energy: energy.BitstringEnergy):
  """Returns the probabilities of the EBM over all bitstrings in `itertools.product` order
  (ebm_utils.py:24-36)."""
  all_bitstrings = torch.tensor(list(itertools.product([0, 1], repeat=input_energy.num_bits)),
                                dtype=torch.int8)
  all_energies = input_energy(all_bitstrings)
  energy_exp = torch.exp(-all_energies)
  partition = torch.sum(energy_exp)
  return energy_exp / partition
