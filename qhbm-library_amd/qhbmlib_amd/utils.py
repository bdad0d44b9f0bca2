"""Utilities used across more than one file (reference: qhbmlib/utils.py)."""
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


def unique_bitstrings_with_counts(bitstrings, out_idx=torch.int32):
  """Unique rows in FIRST-OCCURRENCE order with inverse index and counts
  (utils.py:61-78: tf.raw_ops.UniqueWithCountsV2(axis=[0]); order pinned by
  tests/utils_test.py:165-167).  Runs on the host: it is O(B n) integer work on
  the sampler's output, outside the hot path."""
  bits = torch.as_tensor(bitstrings)
  dev, dt = bits.device, bits.dtype
  arr = bits.detach().cpu().numpy()
  if arr.ndim != 2:
    raise ValueError("bitstrings must be 2-D")
  if arr.shape[0] == 0:
    return (bits.clone(), torch.zeros((0,), dtype=out_idx, device=dev),
            torch.zeros((0,), dtype=out_idx, device=dev))
  packed = np.ascontiguousarray(arr.astype(np.uint8))
  keys = packed.view(np.dtype((np.void, packed.shape[1]))).ravel()
  _, first, inverse, counts = np.unique(keys, return_index=True, return_inverse=True,
                                        return_counts=True)
  order = np.argsort(first, kind="stable")          # sorted-unique -> first-occurrence rank
  rank = np.empty_like(order)
  rank[order] = np.arange(order.size)
  y = arr[np.sort(first)]
  idx = rank[inverse.ravel()]
  cnt = counts[order]
  return (torch.as_tensor(y, dtype=dt, device=dev),
          torch.as_tensor(idx, dtype=out_idx, device=dev),
          torch.as_tensor(cnt, dtype=out_idx, device=dev))


def expand_unique_results(y, idx):
  """expanded[i] = y[idx[i]] (utils.py:81-92)."""
  return y.index_select(0, idx.to(torch.long).to(y.device))
