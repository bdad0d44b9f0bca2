"""Worker of tests/test_dist_qnn_gpu.py: one rank of a torch.distributed.run launch.  Evaluates
`AnalyticQuantumInference(..., process_group=True).expectation` + backward on a fixed model and
batch; rank 0 writes values and gradients to the .npz given on the command line.  The test box has
one GPU, so every rank uses cuda:0 and the collectives run over gloo (parallel.py stages CUDA
tensors through the host for that backend); on a multi-GPU node the same code runs over RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qhbm-library_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

from qhbmlib_amd import inference, ir, models  # noqa: E402
from tests.test_host_api import hea_circuit  # noqa: E402


def main():
  import time
  t0 = time.time()
  T = lambda m: print(f"[worker {time.time() - t0:7.2f}s] {m}", flush=True) if os.environ.get("QHBM_TEST_TRACE") else None
  out_path = sys.argv[1]
  backend = os.environ.get("QHBM_TEST_BACKEND", "gloo")
  dist.init_process_group(backend)
  rank, world = dist.get_rank(), dist.get_world_size()
  torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
  n, layers = 14, 2
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(14)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "d"))
  with torch.no_grad():
    circ.trainable_variables[0].copy_(torch.as_tensor(rng.uniform(-1, 1, len(circ.symbol_names)), dtype=torch.float32))
  xxz = ir.PauliSum()
  for a, b in zip(qubits, qubits[1:]):
    xxz += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
  zsum = ir.PauliSum.from_pauli_strings([ir.PZ(q) for q in qubits])
  uniq = rng.integers(0, 2, size=(11, n)).astype(np.int8)
  states = torch.from_numpy(np.concatenate([uniq, uniq[[3, 3, 7]]]))     # duplicates, as EBM samples have
  weights = torch.from_numpy(rng.normal(size=(states.shape[0], 2)).astype(np.float32))
  T("model built")
  qnn = inference.AnalyticQuantumInference(circ, process_group=True,
                                                   ordered_reduction=os.environ.get("QHBM_TEST_ORDERED", "1") == "1")
  out = qnn.expectation(states, [xxz, zsum])
  T("forward done")
  (out * weights.to(out.device)).sum().backward()
  T("backward done")
  if rank == 0:
    np.savez(out_path, world=world, values=out.detach().cpu().numpy(),
             grad=circ.trainable_variables[0].grad.cpu().numpy())
  T("saved")
  dist.barrier()
  T("barrier")
  dist.destroy_process_group()
  T("destroyed")


if __name__ == "__main__":
  main()
