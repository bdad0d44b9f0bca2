"""Developer tool (GPU): where the EAGER host mirror spends a VQT step at BASELINE configs[1] (12 qubits, depth 8, 1024
samples, fixed multiset): cProfile of `vqt()` + `backward()`, top cumulative entries.   python scripts/experiments/profile_mirror_c2.py [c1|c2]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import torch
from qhbmlib_amd import inference, ir, models, utils
from tests.test_host_api import hea_circuit

n, layers, samples = (4, 2, 32) if (len(sys.argv) > 1 and sys.argv[1] == "c1") else (12, 8, 1024)
qubits = ir.GridQubit.rect(1, n)
torch.manual_seed(0)
circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "p"), tfq_compat_bit_order=False).to("cuda")
energy = models.BernoulliEnergy(list(range(n))).to("cuda")
with torch.no_grad():
  circuit.trainable_variables[0].uniform_(-1, 1); energy.post_process[0].kernel.uniform_(-0.1, 0.1)
e_inf = inference.BernoulliEnergyInference(energy, samples, initial_seed=7)
qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit))
ham = ir.PauliSum()
for i, q in enumerate(qubits):
  ham += -1.0 * ir.PX(q)
  ham += -1.0 * ir.PZ(q) * ir.PZ(qubits[(i + 1) % n])
variables = list(energy.parameters()) + circuit.trainable_variables
with torch.no_grad():
  drawn = e_inf.sample(samples).cuda()
rows, _, counts = utils.unique_bitstrings_with_counts(drawn)

def step(fixed=True):
  for v in variables:
    v.grad = None
  if fixed:
    with e_inf.fixed_samples(rows, counts):
      loss = inference.vqt(qhbm, [ham], 1.0)
      loss.backward()
  else:
    loss = inference.vqt(qhbm, [ham], 1.0)
    loss.backward()
  return loss

for fixed in (True, False):
  for _ in range(10): step(fixed)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(50): step(fixed)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
  print(f"fixed multiset={fixed}: {dt * 1e3:.3f} ms per step (sampler {'excluded' if fixed else 'included: host Bernoulli draw + device dedup'})")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step(True)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
