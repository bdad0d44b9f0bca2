"""Developer tool (GPU): wall time of one host-mirror `vqt()` + `backward()` step with the QAIA ansatz of the reference's
baselines (circuit.py:211-292: layers of exp(-i eta H_zz) exp(-i gamma H_x) built by tfq.util.exponential -- CNOT
ladders around rz, H around rz) on a TFIM ring:  python scripts/qaia_mirror_time.py [samples] [qubits] [layers]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import torch
from qhbmlib_amd import inference, ir, models

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
qs = ir.GridQubit.rect(1, n)
zz, xs = ir.PauliSum(), ir.PauliSum()
for a, b in zip(qs, qs[1:] + qs[:1]):
  zz += ir.PZ(a) * ir.PZ(b)
for q in qs:
  xs += ir.PX(q)
qaia = models.QAIA([zz, xs], [ir.PZ(q) for q in qs], layers)
torch.manual_seed(0)
for v in qaia.value_layers_inputs[0]:
  with torch.no_grad():
    v.uniform_(-0.5, 0.5)
ebm = models.BernoulliEnergy(list(range(n))).to("cuda")
e_inf = inference.BernoulliEnergyInference(ebm, samples, initial_seed=7)
qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(qaia))
tfim = -1.0 * zz - 1.0 * xs
for step in range(4):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  loss = inference.vqt(qhbm, [tfim], 1.0)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  loss.backward()
  torch.cuda.synchronize(); t2 = time.perf_counter()
  (eng,) = list(qhbm.q_inference._engines._engines.values())
  print(f"step {step}: vqt() {t1 - t0:.3f} s + backward {t2 - t1:.3f} s = {t2 - t0:.3f} s  (loss {float(loss.detach()):.5f}, passes {eng.num_passes()})", flush=True)
