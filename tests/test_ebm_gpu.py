"""EBM side on the GPU: qhbm_parity_energy / _vjp against the torch layers of the
host mirror (which follow qhbmlib/models/energy.py:123-209) and the numpy oracle, and
AnalyticEnergyInference with the energy resident on the device."""
import itertools
import time

import numpy as np
import pytest
import torch

from oracle import qhbm_oracle as O
from qhbmlib_amd import inference, ir, models
from tests.test_host_api import hea_circuit

pytestmark = pytest.mark.gpu


def _set(param, values):
  with torch.no_grad():
    param.copy_(torch.as_tensor(np.asarray(values), dtype=torch.float32))


@pytest.mark.parametrize("n,order", [(1, 1), (5, 2), (12, 3), (20, 2), (40, 2), (64, 1)])  # 64: column 63 is the sign bit of the int64 mask
def test_kobe_energy_kernel_matches_layers_and_oracle(n, order):
  rng = np.random.default_rng(n)
  cpu = models.KOBE(list(range(n)), order)
  thetas = rng.uniform(-1, 1, cpu.post_process[0].kernel.numel())
  _set(cpu.post_process[0].kernel, thetas)
  gpu = models.KOBE(list(range(n)), order)
  _set(gpu.post_process[0].kernel, thetas)
  gpu = gpu.to("cuda")
  bits = torch.from_numpy(rng.integers(0, 2, size=(777, n)).astype(np.int8))
  want = cpu(bits)
  got = gpu(bits.cuda())
  assert got.is_cuda and got.shape == (777,)
  np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), atol=2e-5 * max(1, len(thetas))**0.5)
  np.testing.assert_allclose(got.detach().cpu().numpy(), O.kobe_energy(bits.numpy(), thetas, order), atol=1e-4)
  # gradient with respect to theta: sum_i w_i parity_k(x_i)
  w = torch.from_numpy(rng.normal(size=777).astype(np.float32))
  (want_g,) = torch.autograd.grad((want * w).sum(), cpu.post_process[0].kernel)
  (got_g,) = torch.autograd.grad((got * w.cuda()).sum(), gpu.post_process[0].kernel)
  np.testing.assert_allclose(got_g.cpu().numpy(), want_g.numpy(), atol=2e-4)
  # batched leading dimensions, empty batch
  assert gpu(bits.cuda().reshape(7, 111, n)).shape == (7, 111)
  assert gpu(bits[:0].cuda()).shape == (0,)


def test_bernoulli_energy_kernel():
  n = 9
  rng = np.random.default_rng(1)
  e = models.BernoulliEnergy(list(range(n)))
  thetas = rng.uniform(-2, 2, n)
  _set(e.post_process[0].kernel, thetas)
  e = e.to("cuda")
  bits = torch.tensor(list(itertools.product([0, 1], repeat=n)), dtype=torch.int8)
  got = e(bits.cuda()).detach().cpu().numpy()
  np.testing.assert_allclose(got, O.bernoulli_energy(bits.numpy(), thetas), atol=1e-5)
  # float inputs (the Gibbs-with-gradients kernel differentiates w.r.t. x) keep the layer path
  x = bits[:4].to(torch.float32).cuda().requires_grad_(True)
  e(x).sum().backward()
  assert x.grad is not None


def test_analytic_inference_on_device_matches_host():
  """ebm_test.py:515-559 quantities (log partition, entropy) and the sampler, with the energy on the GPU."""
  n = 10
  rng = np.random.default_rng(2)
  thetas = rng.uniform(-1, 1, n + n * (n - 1) // 2)
  host = models.KOBE(list(range(n)), 2)
  _set(host.post_process[0].kernel, thetas)
  dev = models.KOBE(list(range(n)), 2)
  _set(dev.post_process[0].kernel, thetas)
  dev = dev.to("cuda")
  inf_h = inference.AnalyticEnergyInference(host, 1000, initial_seed=3)
  inf_d = inference.AnalyticEnergyInference(dev, 1000, initial_seed=3)
  energy_fn = lambda b: O.kobe_energy(b, thetas, 2)
  np.testing.assert_allclose(float(inf_d.log_partition().detach()), O.log_partition_exact(energy_fn, n), rtol=1e-5)
  np.testing.assert_allclose(float(inf_d.entropy().detach()), O.entropy_exact(energy_fn, n), rtol=1e-4)
  np.testing.assert_allclose(float(inf_d.entropy().detach()), float(inf_h.entropy().detach()), rtol=1e-4)
  # entropy gradient through the HIP VJP equals the host layers' gradient
  (g_d,) = torch.autograd.grad(inf_d.entropy(), dev.post_process[0].kernel)
  (g_h,) = torch.autograd.grad(inf_h.entropy(), host.post_process[0].kernel)
  np.testing.assert_allclose(g_d.cpu().numpy(), g_h.numpy(), atol=2e-5)
  samples = inf_d.sample(200000)
  assert samples.is_cuda and samples.shape == (200000, n)
  p1 = np.exp(-energy_fn(O.all_bitstrings(n)))
  p1 /= p1.sum()
  marg = (p1[:, None] * O.all_bitstrings(n)).sum(0)
  np.testing.assert_allclose(samples.float().mean(0).cpu().numpy(), marg, atol=5 * 0.5 / np.sqrt(200000))
  assert inf_d.distribution.logits.shape == (2**n,)


def test_vqt_step_end_to_end_on_device_n20():
  """BASELINE config 3's model at one GPU's share: KOBE-2 EBM over 20 bits on the device, HEA
  depth 16, XXZ target; one VQT loss + backward.  The EBM side (2^20 energies, log Z, sampling)
  must not dominate: the reference-style host path takes ~8 s for it (measured)."""
  n, layers, samples = 20, 16, 512
  qubits = ir.GridQubit.rect(1, n)
  ebm = models.KOBE(list(range(n)), 2).to("cuda")
  with torch.no_grad():
    ebm.post_process[0].kernel.uniform_(-0.1, 0.1)   # high entropy: mostly distinct samples
  circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "v"))
  e_inf = inference.AnalyticEnergyInference(ebm, samples, initial_seed=7)
  qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit))
  xxz = ir.PauliSum()
  for a, b in zip(qubits, qubits[1:]):
    xxz += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  loss = inference.vqt(qhbm, [xxz], 1.0)
  loss.backward()
  torch.cuda.synchronize()
  first = time.perf_counter() - t0
  g_theta = ebm.post_process[0].kernel.grad
  g_phi = circuit.trainable_variables[0].grad
  assert torch.isfinite(loss) and g_theta.shape == (210,) and g_phi.shape == (944,)
  assert float(g_theta.abs().max()) > 0 and float(g_phi.abs().max()) > 0
  # second step after a variable update: re-runs _ready_inference (2^20 energies) on the device
  with torch.no_grad():
    ebm.post_process[0].kernel.add_(0.01 * torch.randn_like(ebm.post_process[0].kernel))
  t0 = time.perf_counter()
  loss2 = inference.vqt(qhbm, [xxz], 1.0)
  loss2.backward()
  torch.cuda.synchronize()
  second = time.perf_counter() - t0
  print(f"vqt n=20 L=16 {samples} samples: first step {first:.3f} s, next step {second:.3f} s")
  assert second < 3.0


def test_vqt_step_n20_loss_and_both_gradients_against_oracle():
  """The same model (KOBE-2 over 20 bits, HEA depth 16, XXZ) on a 40-sample draw, compared with the
  oracle's VQT loss and gradients on the drawn multiset (vqt_loss.py:46-55 with ebm.py:262-329):
  <H> and d/dphi from the C fp32 restatement (one state per host thread), the EBM side in numpy.
  Tolerances: loss 5e-5 * (beta * sum|c_k| + 1); d/dphi 1e-4 * max(1, |grad|_inf); d/dtheta 2e-4."""
  from oracle import qhbm_cpu as C
  n, layers, samples, beta = 20, 16, 40, 0.7
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(20)
  ebm = models.KOBE(list(range(n)), 2)
  thetas = rng.uniform(-0.1, 0.1, ebm.post_process[0].kernel.numel())
  _set(ebm.post_process[0].kernel, thetas)
  ebm = ebm.to("cuda")
  circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "v"))
  phi = rng.uniform(-1, 1, len(circuit.symbol_names)).astype(np.float32)
  _set(circuit.trainable_variables[0], phi)
  e_inf = inference.AnalyticEnergyInference(ebm, samples, initial_seed=11)
  qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit))
  xxz = ir.PauliSum()
  for a, b in zip(qubits, qubits[1:]):
    xxz += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
  loss = inference.vqt(qhbm, [xxz], beta)
  loss.backward()
  drawn = e_inf.sample(samples).cpu().numpy()   # fixed seed: the draw vqt() used
  g_theta = ebm.post_process[0].kernel.grad.cpu().numpy()
  g_phi = circuit.trainable_variables[0].grad.cpu().numpy()
  # ---- oracle on the drawn multiset ----
  uniq, _, counts = O.unique_bitstrings_with_counts(drawn)
  gates = circuit.pqc.flat_gates(qubits, circuit.symbol_names)
  op = xxz.masks(qubits)
  weights = counts / counts.sum()
  h, want_phi = C.expectation_vjp(n, gates, phi, uniq, [op], (beta * weights)[:, None].astype(np.float32))
  index_sets = O.parity_indices(n, 2)
  feats = O.parities(uniq, index_sets)                       # [U, 210] = dE/dtheta
  f = beta * h[:, 0].astype(np.float64) - feats @ thetas
  log_z, chunk = -np.inf, 1 << 16
  for lo in range(0, 1 << n, chunk):                         # exact log Z over all 2^20 bitstrings
    idx = np.arange(lo, lo + chunk)
    bits = ((idx[:, None] >> np.arange(n - 1, -1, -1)[None, :]) & 1).astype(np.int8)
    log_z = np.logaddexp(log_z, np.logaddexp.reduce(-(O.parities(bits, index_sets) @ thetas)))
  want_loss = float(weights @ f - log_z)
  avg_f = float(weights @ f)
  want_theta = (weights @ feats) * avg_f - weights @ (feats * f[:, None])
  norm = sum(abs(c) for c, _, _ in op)
  assert abs(float(loss.detach()) - want_loss) <= 5e-5 * (beta * norm + 1.0), (float(loss.detach()), want_loss)
  np.testing.assert_allclose(g_phi, want_phi, atol=1e-4 * max(1.0, np.abs(want_phi).max()), rtol=0)
  np.testing.assert_allclose(g_theta, want_theta, atol=2e-4, rtol=0)


def test_gibbs_with_gradients_with_the_energy_on_the_device():
  """ebm_test.py:879-947 with the EBM resident on the GPU: the chain's energy differences go through
  qhbm_parity_energy (integer CUDA bitstrings), its index proposals through the torch layers on the
  device; the sampled distribution must match the exact one.  6000 correlated samples: entropy within
  rtol 8e-2, probabilities within 4e-2 (the same chain on the CPU lands 5.5 % / 2.2e-2 off with this
  seed and 1.5 % / 5e-3 off after 30000 samples), every bitstring visited."""
  num_bits = 4
  rng = np.random.default_rng(4)
  energy = models.KOBE(list(range(num_bits)), 2)
  thetas = rng.uniform(-0.8, 0.8, energy.post_process[0].kernel.numel())
  _set(energy.post_process[0].kernel, thetas)
  energy = energy.to("cuda")
  n_samples = 6000
  layer = inference.GibbsWithGradientsInference(energy, n_samples, 500, initial_seed=9)
  samples = layer.sample(n_samples)
  assert samples.shape == (n_samples, num_bits) and samples.dtype == torch.int8
  all_bits = O.all_bitstrings(num_bits)
  p = np.exp(-O.kobe_energy(all_bits, thetas, 2))
  p /= p.sum()
  counts = np.zeros(2**num_bits)
  idx = (samples.numpy().astype(np.int64) * (1 << np.arange(num_bits - 1, -1, -1))).sum(1)
  np.add.at(counts, idx, 1)
  q = counts / counts.sum()
  assert (counts > 0).all()
  entropy = lambda d: float(-(d[d > 0] * np.log(d[d > 0])).sum())
  np.testing.assert_allclose(entropy(q), entropy(p), rtol=8e-2)
  np.testing.assert_allclose(q, p, atol=4e-2)
  assert abs(entropy(q) - np.log(2**num_bits)) > 0.2 * np.log(2**num_bits)   # and it is not uniform
