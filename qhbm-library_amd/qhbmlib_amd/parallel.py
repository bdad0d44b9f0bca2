"""Batch sharding of the expectation hot path over the GPUs of one node.

The unit of work is one unique EBM bitstring = one independent statevector that
shares only the parameters (broadcast) and the observable.
Rows are dealt out in contiguous blocks, one process per GPU; a statevector is
never split.  One exchange per call: an all-gather of the per-state expectation
values and an all-reduce (sum) of the [P] gradient -- KiB-sized messages over
RCCL/xGMI (latency-bound), nothing else crosses GPUs.
"""
from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def partition(num_rows: int, world_size: int) -> List[Tuple[int, int]]:
  """Contiguous [lo, hi) block per rank; the first `num_rows % world_size` ranks get
  one extra row."""
  base, extra = divmod(num_rows, world_size)
  out, lo = [], 0
  for r in range(world_size):
    hi = lo + base + (1 if r < extra else 0)
    out.append((lo, hi))
    lo = hi
  return out


class ShardedExpectation:
  """Wraps a single-device `expectation_vjp(bits, params, upstream) -> (vals, grad)`
  (e.g. `Engine.expectation_vjp`) into the same call over the whole process group."""

  def __init__(self, local_expectation_vjp: Callable, group=None):
    self._local = local_expectation_vjp
    self._group = group

  def expectation_vjp(self, bits: torch.Tensor, params: torch.Tensor, upstream: torch.Tensor):
    """bits [U, n], upstream [U, T] are the FULL batch on every rank (tiny, host-made);
    returns (values [U, T], grad [P]) identical on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
      return self._local(bits, params, upstream)
    world = dist.get_world_size(self._group)
    rank = dist.get_rank(self._group)
    blocks = partition(bits.shape[0], world)
    lo, hi = blocks[rank]
    vals_local, grad = self._local(bits[lo:hi], params, upstream[lo:hi])
    n_ops = upstream.shape[1]
    width = max(h - l for l, h in blocks)
    padded = torch.zeros((width, n_ops), dtype=vals_local.dtype, device=vals_local.device)
    padded[:hi - lo] = vals_local
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=self._group)
    dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self._group)
    vals = torch.cat([g[:h - l] for g, (l, h) in zip(gathered, blocks)], 0)
    return vals, grad
