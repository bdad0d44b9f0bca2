"""Developer tool (GPU): wall time of one host-mirror `qmhl()` + `backward()` step at BASELINE config 3's size (data
from a fixed QHBM, model = KOBE-2 EBM over 20 bits on the device + HEA depth 16; the modular Hamiltonian is measured
as 210 Z-string shards after U^dagger):  python scripts/qmhl_mirror_time.py [samples] [qubits] [layers]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import torch
from qhbmlib_amd import data, inference, ir, models
from tests.test_host_api import hea_circuit

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
qubits = ir.GridQubit.rect(1, n)


def make_qhbm(tag, seed):
  ebm = models.KOBE(list(range(n)), 2).to("cuda")
  with torch.no_grad():
    ebm.post_process[0].kernel.uniform_(-0.1, 0.1)
  circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, tag))
  e_inf = inference.AnalyticEnergyInference(ebm, samples, initial_seed=seed)
  return inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit)), ebm, circuit


target, t_ebm, t_circuit = make_qhbm("d", 11)
if os.environ.get("QHBM_FREEZE_DATA", "1") != "0":  # the data source is fixed: nobody wants its gradients (default)
  for p in list(t_ebm.parameters()) + t_circuit.trainable_variables:
    p.requires_grad_(False)
model, ebm, circuit = make_qhbm("m", 7)
qdata = data.QHBMData(target)
for step in range(4):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  loss = inference.qmhl(qdata, model)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  loss.backward()
  torch.cuda.synchronize(); t2 = time.perf_counter()
  print(f"step {step}: qmhl() {t1 - t0:.3f} s + backward {t2 - t1:.3f} s = {t2 - t0:.3f} s  (loss {float(loss.detach()):.5f})", flush=True)
  with torch.no_grad():
    for p in list(ebm.parameters()) + circuit.trainable_variables:
      if p.grad is not None:
        p.add_(-0.01 * p.grad); p.grad = None
if os.environ.get("QHBM_KERNEL_TIMES"):
  for q in (target, model):
    for eng in q.q_inference._engines._engines.values():
      eng.set_option("profile_events", 1)
  loss = inference.qmhl(qdata, model); loss.backward(); torch.cuda.synchronize()
  for name, q in (("data", target), ("model", model)):
    for eng in q.q_inference._engines._engines.values():
      print(name, eng.num_passes(), {k: round(v, 2) for k, v in eng.kernel_time_ms().items()}, flush=True)
