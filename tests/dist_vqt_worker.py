"""Worker of tests/test_dist_vqt_gpu.py: one rank of a torch.distributed.run launch running whole
training steps the way the reference's loop does (sample -> dedup -> hot path, ebm.py:271-280):
`inference.vqt(qhbm, [H], beta)` + `backward()` with `AnalyticEnergyInference(initial_seed=None)` and
`AnalyticQuantumInference(process_group=True)`.  Each rank seeds torch's global generator DIFFERENTLY
(100 + rank): only the seed agreement (`QHBM.agree_seeds`, run lazily by `vqt`; called explicitly here so
that the agreed seed can be recorded before the first step) makes the ranks draw one sample set.
QHBM_TEST_DESYNC=1 gives the samplers explicit, different seeds instead: the run must then fail with
ShardMismatchError on every rank.  One GPU on the test box: every rank uses cuda:0, collectives over gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qhbm-library_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

from qhbmlib_amd import inference, ir, models, parallel  # noqa: E402
from tests.test_host_api import hea_circuit  # noqa: E402


def main():
  out_path = sys.argv[1]
  backend = os.environ.get("QHBM_TEST_BACKEND", "gloo")
  dist.init_process_group(backend)
  rank, world = dist.get_rank(), dist.get_world_size()
  torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
  torch.manual_seed(100 + rank)
  n, layers, samples, beta = 14, 2, 48, 0.7
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(77)
  ebm = models.KOBE(list(range(n)), 2)
  with torch.no_grad():
    ebm.post_process[0].kernel.copy_(torch.as_tensor(rng.uniform(-0.3, 0.3, ebm.post_process[0].kernel.numel()),
                                                     dtype=torch.float32))
  ebm = ebm.to("cuda")
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "w"))
  with torch.no_grad():
    circ.trainable_variables[0].copy_(torch.as_tensor(rng.uniform(-1, 1, len(circ.symbol_names)), dtype=torch.float32))
  desync = os.environ.get("QHBM_TEST_DESYNC") == "1"
  e_inf = inference.AnalyticEnergyInference(ebm, samples, initial_seed=(rank + 1) if desync else None)
  q_inf = inference.AnalyticQuantumInference(circ, process_group=True,
                                                   ordered_reduction=os.environ.get("QHBM_TEST_ORDERED", "1") == "1")
  qhbm = inference.QHBM(e_inf, q_inf)
  xxz = ir.PauliSum()
  for a, b in zip(qubits, qubits[1:]):
    xxz += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
  seed_before = e_inf.seed     # local draw from this rank's global generator (100 + rank): differs between ranks
  qhbm.agree_seeds()
  first_seed = e_inf.seed
  record = {}
  try:
    for step in range(2):      # the second step runs on the seed the first one advanced
      for v in (ebm.post_process[0].kernel, circ.trainable_variables[0]):
        v.grad = None
      loss = inference.vqt(qhbm, [xxz], beta)
      loss.backward()
      record[f"loss{step}"] = loss.detach().cpu().numpy()
      record[f"g_phi{step}"] = circ.trainable_variables[0].grad.cpu().numpy()
      record[f"g_theta{step}"] = ebm.post_process[0].kernel.grad.cpu().numpy()
  except parallel.ShardMismatchError as exc:
    print(f"rank {rank}: ShardMismatchError: {exc}", file=sys.stderr, flush=True)
    dist.destroy_process_group()
    sys.exit(7)
  if rank == 0:
    np.savez(out_path, world=world, first_seed=first_seed, seed_before=seed_before, **record)
  dist.barrier()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
