"""Developer tool (GPU): wall time of one host-mirror `vqt()` + `backward()` step on BASELINE config 3's model
(KOBE-2 EBM over 20 bits on the device, HEA depth 16, XXZ target):  python scripts/vqt_mirror_time.py [samples]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import torch
from qhbmlib_amd import inference, ir, models
from tests.test_host_api import hea_circuit

n, layers, samples = 20, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
qubits = ir.GridQubit.rect(1, n)
ebm = models.KOBE(list(range(n)), 2).to("cuda")
with torch.no_grad():
  ebm.post_process[0].kernel.uniform_(-0.1, 0.1)
circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "v"))
e_inf = inference.AnalyticEnergyInference(ebm, samples, initial_seed=7)
qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit))
xxz = ir.PauliSum()
for a, b in zip(qubits, qubits[1:]):
  xxz += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
for step in range(4):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  loss = inference.vqt(qhbm, [xxz], 1.0)
  t1 = time.perf_counter()
  loss.backward()
  torch.cuda.synchronize(); t2 = time.perf_counter()
  (eng,) = list(qhbm.q_inference._engines._engines.values())
  print(f"step {step}: vqt() {t1 - t0:.3f} s + backward {t2 - t1:.3f} s = {t2 - t0:.3f} s  (loss {float(loss.detach()):.5f}, "
        f"engine workspace {eng.allocated_bytes() / 2**30:.1f} GiB)", flush=True)
  with torch.no_grad():
    for p in list(ebm.parameters()) + circuit.trainable_variables:
      p.add_(-0.01 * p.grad); p.grad = None
