"""Utilities used across more than one file (reference: qhbmlib/utils.py)."""
import weakref

import numpy as np
import torch


class Squeeze(torch.nn.Module):
  """Wraps torch.squeeze in a layer (utils.py:20-40)."""

  def __init__(self, axis=None):
    super().__init__()
    self._axis = axis

  def forward(self, inputs):
    if self._axis is None:
      return torch.squeeze(inputs)
    return torch.squeeze(inputs, self._axis)


def weighted_average(counts: torch.Tensor, values: torch.Tensor):
  """Count-weighted mean over the first axis (utils.py:43-58)."""
  float_counts = counts.to(torch.float32).to(values.device)
  weighted = torch.tensordot(float_counts, values.to(torch.float32), dims=([0], [0]))
  return weighted / float_counts.sum()


_MAX_KEY_BITS = 62   # rows of 0/1 with at most this many columns pack into one int64 key


class _KnownUnique:
  """The last array `unique_bitstrings_with_counts` returned: a caller that hands it straight back -- `EnergyInference.
  _expectation` deduplicates the samples (ebm.py:271-273) and the function it then maps over the unique rows,
  `QuantumInference.expectation`, deduplicates again (qnn.py:66-67) -- gets the identity answer without a second sort."""
  ref = None        # weakref to the tensor
  version = -1      # its in-place modification counter when it was produced


def _remember_unique(y):
  _KnownUnique.ref = weakref.ref(y)
  _KnownUnique.version = y._version  # pylint: disable=protected-access


def mark_rows_unique(bits):
  """Declares `bits` deduplicated already: the next `unique_bitstrings_with_counts(bits)` returns it as it is (until it
  is written to).  For callers that hold a multiset in (rows, counts) form -- `EnergyInference.fixed_samples`."""
  _remember_unique(bits)


def _is_known_unique(bits):
  ref = _KnownUnique.ref
  return ref is not None and ref() is bits and bits._version == _KnownUnique.version  # pylint: disable=protected-access


def _first_occurrence_order(first, inverse, counts, rows, out_idx):
  """Sorted-unique groups (row index of each group's first occurrence, inverse index, counts) -> first-occurrence order."""
  order = torch.argsort(first)                      # groups by where they first occur (all distinct: no ties)
  rank = torch.empty_like(order)
  rank[order] = torch.arange(order.numel(), device=order.device)
  return rows.index_select(0, first[order]), rank[inverse].to(out_idx), counts[order].to(out_idx)


def _unique_on_device(bits, out_idx):
  """The same answer as the host path below from torch ops on the tensor's own device (no copy of the rows to the host;
  the only synchronisation is the one that learns how many unique rows there are).  Rows of 0/1 with <= 62 columns are
  packed into int64 keys and sorted stably; anything else goes through torch.unique(dim=0)."""
  b, n = bits.shape
  positions = torch.arange(b, device=bits.device)
  packable = n <= _MAX_KEY_BITS and not bits.is_floating_point() and not bits.is_complex()
  if packable:
    wide = bits.to(torch.int64)
    weights = torch.ones((), dtype=torch.int64, device=bits.device) << torch.arange(n - 1, -1, -1, device=bits.device)
    keys = (wide * weights).sum(1)
    binary = ((wide == 0) | (wide == 1)).all()
    skeys, order = torch.sort(keys, stable=True)    # equal keys keep their input order: a group's first element is its first occurrence
    new = torch.ones(b, dtype=torch.bool, device=bits.device)
    new[1:] = skeys[1:] != skeys[:-1]
    if bool(binary):                                # (the one synchronisation; the mask indexing below would wait anyway)
      group = torch.cumsum(new, 0) - 1
      first = order[new]
      counts = torch.bincount(group, minlength=first.numel())
      inverse = torch.empty_like(group)
      inverse[order] = group
      return _first_occurrence_order(first, inverse, counts, bits, out_idx)
  _, inverse, counts = torch.unique(bits, dim=0, return_inverse=True, return_counts=True)
  first = torch.full((counts.numel(),), b, dtype=torch.int64, device=bits.device)
  first.scatter_reduce_(0, inverse, positions, reduce="amin")
  return _first_occurrence_order(first, inverse, counts, bits, out_idx)


def unique_bitstrings_with_counts(bitstrings, out_idx=torch.int32):
  """Unique rows in FIRST-OCCURRENCE order with inverse index and counts
  (utils.py:61-78: tf.raw_ops.UniqueWithCountsV2(axis=[0]); order pinned by
  tests/utils_test.py:165-167).  CUDA tensors are deduplicated on the device (`_unique_on_device`: bit-identical to the
  host path, tests/test_host_api.py), host tensors with numpy; an array this function just returned is recognised and
  not sorted again."""
  bits = torch.as_tensor(bitstrings)
  dev, dt = bits.device, bits.dtype
  if bits.dim() != 2:
    raise ValueError("bitstrings must be 2-D")
  if bits.shape[0] == 0:
    return (bits.clone(), torch.zeros((0,), dtype=out_idx, device=dev),
            torch.zeros((0,), dtype=out_idx, device=dev))
  if _is_known_unique(bits):
    u = bits.shape[0]
    return bits, torch.arange(u, dtype=out_idx, device=dev), torch.ones((u,), dtype=out_idx, device=dev)
  if bits.is_cuda:
    y, idx, cnt = _unique_on_device(bits.detach(), out_idx)
    _remember_unique(y)
    return y, idx, cnt
  arr = bits.detach().numpy()
  packed = np.ascontiguousarray(arr.astype(np.uint8))
  keys = packed.view(np.dtype((np.void, packed.shape[1]))).ravel()
  _, first, inverse, counts = np.unique(keys, return_index=True, return_inverse=True,
                                        return_counts=True)
  order = np.argsort(first, kind="stable")          # sorted-unique -> first-occurrence rank
  rank = np.empty_like(order)
  rank[order] = np.arange(order.size)
  y = torch.as_tensor(arr[np.sort(first)], dtype=dt, device=dev)
  idx = rank[inverse.ravel()]
  cnt = counts[order]
  _remember_unique(y)
  return (y, torch.as_tensor(idx, dtype=out_idx, device=dev),
          torch.as_tensor(cnt, dtype=out_idx, device=dev))


def expand_unique_results(y, idx):
  """expanded[i] = y[idx[i]] (utils.py:81-92)."""
  return y.index_select(0, idx.to(torch.long).to(y.device))
