import os, sys, time, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import torch
from qhbmlib_amd import inference, ir, models
from tests.test_host_api import hea_circuit
n, layers, samples = 20, 16, int(sys.argv[1])
qubits = ir.GridQubit.rect(1, n)
ebm = models.KOBE(list(range(n)), 2).to("cuda")
with torch.no_grad(): ebm.post_process[0].kernel.uniform_(-0.1, 0.1)
circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "v"))
e_inf = inference.AnalyticEnergyInference(ebm, samples, initial_seed=7)
qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit))
xxz = ir.PauliSum()
for a, b in zip(qubits, qubits[1:]): xxz += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
def step():
  loss = inference.vqt(qhbm, [xxz], 1.0); loss.backward(); torch.cuda.synchronize()
for _ in range(2): step()
# engine-only time
(eng,) = list(qhbm.q_inference._engines._engines.values())
eng.set_option("profile_events", 1); step(); kt = eng.kernel_time_ms(); eng.set_option("profile_events", 0)
t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
print(f"step {dt*1e3:.1f} ms, engine kernels {kt['fwd_ms']+kt['bwd_ms']+kt['obs_ms']:.1f} ms")
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
