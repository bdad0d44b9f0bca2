"""PROJECTED 8-GPU step of BASELINE config 3 from what ONE GPU can measure (VERDICT r4 #5i) -- not a measurement.

No 8-GPU node is available to this repository's own runs (the driver's round-end scaling run is the only source of a
real curve).  What one box can measure: (a) the step of ONE rank's shard -- 4096 / 8 = 512 states of config 3 -- with
the exact engine calls `bench.py --gpus 8` issues per rank, and (b) the latency of the two collectives of the step
(all-gather of the [U_r, 1] values, all-reduce of the [P] gradient) through RCCL with a world of ONE rank (the RCCL
code path, device tensors, no xGMI hop).  projected step = (a) + (b); a real run adds the xGMI hops of a
latency-bound 3.7 KiB ring and any launch skew between the ranks, so the projection is an UPPER bound on throughput.

  gpurun -- 'python3 scripts/projection_8gpu.py > gpurun_out/r05/projection_8gpu.json'
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from qhbmlib_amd import _engine as E  # noqa: E402


def main():
  n, layers, total, world = 20, 16, 4096, 8
  shard = total // world
  steps, warmup = 20, 3
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("MASTER_PORT", "29541")
  torch.cuda.set_device(0)
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
  gates, n_params = bench.hea_gates(n, layers)
  eng = E.Engine(0)
  eng.set_circuit(n, gates, n_params)
  eng.set_observables([bench.xxz_op(n)])
  bits = torch.from_numpy(bench.distinct_bitstrings(n, total, 4321)[:shard]).cuda()
  params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, n_params).astype(np.float32)).cuda()
  up = torch.full((shard, 1), 1.0 / total, device="cuda")
  gathered = [torch.empty((shard, 1), device="cuda")]

  def step(with_exchange):
    vals, grad = eng.expectation_vjp(bits, params, up)
    if with_exchange:
      dist.all_reduce(grad)
      dist.all_gather(gathered, vals)
    return vals, grad

  out = {}
  for tag, ex in (("engine_only", False), ("with_world1_rccl_exchange", True)):
    for _ in range(warmup):
      step(ex)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      step(ex)
    torch.cuda.synchronize()
    out[tag] = (time.perf_counter() - t0) / steps * 1e3
  # the two collectives alone, back to back, stream-ordered
  g = torch.zeros(n_params, device="cuda")
  v = torch.zeros((shard, 1), device="cuda")
  for _ in range(20):
    dist.all_reduce(g)
    dist.all_gather(gathered, v)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  iters = 500
  for _ in range(iters):
    dist.all_reduce(g)
    dist.all_gather(gathered, v)
  torch.cuda.synchronize()
  pair_us = (time.perf_counter() - t0) / iters * 1e6
  dist.destroy_process_group()
  head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
  step_ms = out["with_world1_rccl_exchange"]
  print(json.dumps({
      "label": "PROJECTED, NOT MEASURED: one MI355X running one rank's shard; no 8-GPU node was available",
      "workload": "BASELINE config 3: 20 qubits, depth 16, XXZ, 4096 states over 8 ranks = 512 states per rank",
      "git_head": head or None,
      "measured_on_one_gpu": {
          "shard_states": shard, "steps": steps,
          "step_ms_engine_only": out["engine_only"],
          "step_ms_with_world1_rccl_allreduce_and_allgather": step_ms,
          "rccl_world1_allreduce_P_plus_allgather_values_us": pair_us,
          "exchange_bytes_per_rank": 4 * n_params + 4 * total,
      },
      "projected_8gpu": {
          "step_ms": step_ms,
          "evals_per_s": total * 57 / (step_ms * 1e-3),
          "assumes": "ranks run in lock step; the xGMI hops of the 3.7 KiB all-reduce and the 16 KiB all-gather "
                     "(7 ring steps each, latency-bound) are not included: add ~10-50 us per step on a real node",
      },
  }, indent=1))


if __name__ == "__main__":
  main()
