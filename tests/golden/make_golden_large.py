"""Golden vectors at the BASELINE.json sizes (n = 20 / 24 / 28), generated in the build
container -- the reference cannot run here and the oracle is too slow to run
at these sizes inside the GPU tests, so its outputs are committed as fixtures:

    python tests/golden/make_golden_large.py c3      # ~10 min, 4 processes
    python tests/golden/make_golden_large.py c4      # ~10 min
    python tests/golden/make_golden_large.py c5      # ~10 min, one thread, 4.5 GiB
    python tests/golden/make_golden_large.py c4qmhl  # ~10 min, 6 threads
    python tests/golden/make_golden_large.py c5vjp   # ~25 min, 3 threads, 13 GiB
    python tests/golden/make_golden_large.py c4d16   # ~15 min, one state

  c3_n20_l16.npz  BASELINE config 3's circuit: 20 qubits, HEA depth 16 (944 parameters), XXZ
                  chain.  4 states: values and per-state gradient rows by the numpy complex128
                  oracle (oracle/qhbm_oracle.py), cross-checked here against the independent C
                  fp32 restatement (oracle/qhbm_cpu.c) on every state; plus the 210 KOBE-2 shard
                  values of one state through the modular-Hamiltonian circuit U(phi) V(phi_h)^dagger
                  (1888 gates).
  c4_n24_d2.npz   24 qubits, HEA depth 2, the random 512-term Pauli sum of config 4: all 512 term
                  values and the VJP of their sum for one state (C oracle; 8 terms cross-checked
                  against the numpy oracle).
  c5_n28_d2.npz   28 qubits (2 GiB per state), HEA depth 2, TFIM ring: the 56 term values of one
                  state (C oracle, whose same code is cross-checked at n = 20 and n = 24 above).

  c4_qmhl_n24_d4.npz  BASELINE config 4 as a LOSS: QMHL (qmhl_loss.py:33-34) at 24 qubits -- data circuit
                  (HEA depth 4) on 6 distinct bitstrings with multiplicities, model = Bernoulli energy +
                  HEA depth 4: the loss <H_model>_data + log Z and its gradients with respect to the
                  model's thetas and circuit parameters, from the C oracle's adjoint VJP over the
                  568-gate circuit U_data V(phi)^dagger (ebm side in closed form).
  c5_n28_d2_vjp.npz   config 5's width with a BATCH: three 2 GiB states at 28 qubits, HEA depth 2, TFIM
                  ring: the 56 term values of every state and the [P] VJP of the weighted sum
                  (C oracle adjoint) -- what a streamed (chunk_states = 1) engine run is compared with.

  c4_n24_d16_shift.npz  config 4 at its stated depth 16 (1136 parameters), 24 of the 512 terms: term values and
                  the gradient of their sum (C oracle adjoint) for one state -- the reference for the engine's
                  parameter-shift gradient at config 4's literal size.

Inputs follow SURVEY.md 8(d): phi ~ U[-1, 1] from a fixed seed, seeded bitstrings.  Gate lists are
[G, 6] = (kind, q0, q1, param_idx, scalar, offset); ops are [T, 4] = (op, coeff, x_mask, z_mask).
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import qhbm_cpu as C  # noqa: E402
from oracle import qhbm_oracle as O  # noqa: E402
from tests.golden.make_golden import pack_ops, save  # noqa: E402


def _c3_state(args):
  n, gates, params, bits, op = args
  vals, jac = O.expectation_jacobian(n, gates, params, bits[None, :], [op])
  return vals[0, 0], jac[0, 0]


def make_c3():
  n, layers = 20, 16
  rng = np.random.default_rng(2016)
  gates, names = O.hea_gates(n, layers, "b")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(4, n)).astype(np.int8)
  bits[0] = 0                      # |0...0>
  bits[1] = 1                      # |1...1>: every tile but the last is zero after the first pass
  op = O.xxz_chain_op(n)
  t0 = time.time()
  with mp.Pool(4) as pool:
    res = pool.map(_c3_state, [(n, gates, params, b, op) for b in bits])
  values = np.array([r[0] for r in res])
  grads = np.stack([r[1] for r in res])
  print(f"c3 numpy oracle: {time.time() - t0:.0f} s; values", values)
  # cross-check: the C restatement, one VJP per state with unit upstream
  for s in range(4):
    cv, cg = C.expectation_vjp(n, gates, params, bits[s:s + 1], [op], np.ones((1, 1), np.float32))
    assert abs(cv[0, 0] - values[s]) < 5e-5 * sum(abs(c) for c, _, _ in op), (s, cv, values[s])
    assert np.abs(cg - grads[s]).max() < 1e-4 * max(1.0, np.abs(grads[s]).max()), np.abs(cg - grads[s]).max()
  print("c3: C oracle agrees on all 4 states")
  # modular Hamiltonian: bit . U(phi) . V(phi_h)^dagger, KOBE-2 shards, state 2
  v_gates, v_names = O.hea_gates(n, layers, "h")
  p_u = len(names)
  v_shift = [(k, q0, q1, p + p_u, s, o) for (k, q0, q1, p, s, o) in v_gates]
  v_params = rng.uniform(-1, 1, len(v_names))
  total = gates + O.inverse_gates(v_shift)
  all_params = np.concatenate([params, v_params])
  shards = O.kobe_shards(n, 2)
  t0 = time.time()
  psi = O.simulate(n, total, all_params, bits[2])
  shard_values = np.array([O.op_expectation(psi, sh) for sh in shards])
  print(f"c3 shards: {time.time() - t0:.0f} s")
  cs = C.expectation(n, total, all_params, bits[2:3], shards)
  assert np.abs(cs[0] - shard_values).max() < 5e-5, np.abs(cs[0] - shard_values).max()
  save("c3_n20_l16.npz", n=n, layers=layers, gates=np.array(gates, dtype=np.float64), params=params, bits=bits,
       ops=pack_ops([op]), values=values, grads=grads, total_gates=np.array(total, dtype=np.float64),
       total_params=all_params, kobe2_shards=pack_ops(shards), kobe2_state=2, kobe2_values=shard_values)


def make_c4():
  n, layers = 24, 2
  rng = np.random.default_rng(2402)
  gates, names = O.hea_gates(n, layers, "b")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(1, n)).astype(np.int8)
  op = O.random_pauli_op(n, 512, 24)
  per_term = [[t] for t in op]
  t0 = time.time()
  term_values = C.expectation(n, gates, params, bits, per_term)[0].astype(np.float64)
  _, grad = C.expectation_vjp(n, gates, params, bits, [op], np.ones((1, 1), np.float32))
  print(f"c4 C oracle: {time.time() - t0:.0f} s; sum {term_values.sum():.6f}")
  t0 = time.time()
  psi = O.simulate(n, gates, params, bits[0])
  picks = [0, 1, 17, 100, 255, 256, 400, 511]
  for k in picks:
    want = O.op_expectation(psi, per_term[k])
    assert abs(want - term_values[k]) < 2e-5 * max(1.0, abs(op[k][0])), (k, want, term_values[k])
    term_values[k] = want
  print(f"c4 numpy cross-check of {len(picks)} terms: {time.time() - t0:.0f} s")
  save("c4_n24_d2.npz", n=n, layers=layers, gates=np.array(gates, dtype=np.float64), params=params, bits=bits,
       ops=pack_ops([op]), term_values=term_values, grad=grad.astype(np.float64), numpy_checked=np.array(picks))


def make_c5():
  n, layers = 28, 2
  rng = np.random.default_rng(2802)
  gates, names = O.hea_gates(n, layers, "b")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(1, n)).astype(np.int8)
  op = O.tfim_ring_op(n)
  t0 = time.time()
  term_values = C.expectation(n, gates, params, bits, [[t] for t in op], n_threads=1)[0].astype(np.float64)
  print(f"c5 C oracle: {time.time() - t0:.0f} s; sum {term_values.sum():.6f}")
  save("c5_n28_d2.npz", n=n, layers=layers, gates=np.array(gates, dtype=np.float64), params=params, bits=bits,
       ops=pack_ops([op]), term_values=term_values)


def make_c4_qmhl():
  n, layers = 24, 4
  rng = np.random.default_rng(2404)
  d_gates, d_names = O.hea_gates(n, layers, "d")       # data circuit, fixed values
  m_gates, m_names = O.hea_gates(n, layers, "m")       # model circuit V(phi)
  p_d = len(d_names)
  d_params = rng.uniform(-1, 1, p_d)
  m_params = rng.uniform(-1, 1, len(m_names))
  thetas = rng.uniform(-1, 1, n)                       # Bernoulli energy E(x) = sum_k theta_k (1 - 2 x_k)
  uniq = rng.integers(0, 2, size=(6, n)).astype(np.int8)
  counts = np.array([3, 1, 2, 1, 1, 2])
  samples = np.repeat(uniq, counts, axis=0)[rng.permutation(counts.sum())]
  u2, _, c2 = O.unique_bitstrings_with_counts(samples)   # first-occurrence order of the shuffled samples
  weights = c2 / c2.sum()
  m_shift = [(k, q0, q1, p + p_d, s, o) for (k, q0, q1, p, s, o) in m_gates]
  total = d_gates + O.inverse_gates(m_shift)            # data circuit, then V(phi)^dagger (qnn.py:69-72)
  all_params = np.concatenate([d_params, m_params])
  shards = O.bernoulli_shards(n)                        # Z_k
  h_op = [(float(thetas[k]) * c, x, z) for k, sh in enumerate(shards) for (c, x, z) in sh]
  t0 = time.time()
  vals, grad = C.expectation_vjp(n, total, all_params, u2, [h_op], weights[:, None].astype(np.float32), n_threads=6)
  print(f"c4 qmhl C adjoint: {time.time() - t0:.0f} s")
  t0 = time.time()
  shard_vals = C.expectation(n, total, all_params, u2, shards, n_threads=6).astype(np.float64)   # [U, n]
  print(f"c4 qmhl C shards: {time.time() - t0:.0f} s")
  assert np.abs(shard_vals @ thetas - vals[:, 0]).max() < 2e-5 * np.abs(thetas).sum()
  log_z = float(np.sum(np.log(2.0 * np.cosh(thetas))))
  loss = float(weights @ vals[:, 0].astype(np.float64)) + log_z
  # d loss / d theta_k = <Z_k>_data + d log Z / d theta_k;  log Z = sum log(e^theta + e^-theta)
  g_theta = weights @ shard_vals + np.tanh(thetas)
  save("c4_qmhl_n24_d4.npz", n=n, layers=layers, data_gates=np.array(d_gates, dtype=np.float64), data_params=d_params,
       model_gates=np.array(m_gates, dtype=np.float64), model_params=m_params, thetas=thetas, samples=samples,
       total_gates=np.array(total, dtype=np.float64), loss=loss, log_partition=log_z,
       grad_thetas=g_theta, grad_model_params=grad[p_d:].astype(np.float64), grad_data_params=grad[:p_d].astype(np.float64),
       state_values=vals[:, 0].astype(np.float64), shard_values=shard_vals)


def make_c5_vjp():
  n, layers = 28, 2
  rng = np.random.default_rng(2803)
  gates, names = O.hea_gates(n, layers, "b")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  op = O.tfim_ring_op(n)
  up = np.array([[0.6], [-0.3], [0.9]], np.float32)
  t0 = time.time()
  vals, grad = C.expectation_vjp(n, gates, params, bits, [op], up, n_threads=3)
  print(f"c5 vjp C adjoint: {time.time() - t0:.0f} s; values", vals[:, 0])
  t0 = time.time()
  term_values = C.expectation(n, gates, params, bits, [[t] for t in op], n_threads=3).astype(np.float64)
  print(f"c5 vjp C terms: {time.time() - t0:.0f} s")
  assert np.abs(term_values.sum(1) - vals[:, 0]).max() < 5e-5 * 56
  save("c5_n28_d2_vjp.npz", n=n, layers=layers, gates=np.array(gates, dtype=np.float64), params=params, bits=bits,
       ops=pack_ops([op]), term_values=term_values, upstream=up, grad=grad.astype(np.float64))


def make_c4_d16():
  """BASELINE config 4 at its STATED depth: 24 qubits, HEA depth 16 (1136 parameters), the first 24 terms of the
  random 512-term Pauli sum plus their sum: term values and the [P] gradient of the sum for one state, by the C
  oracle's adjoint -- what the engine's PARAMETER-SHIFT gradient at this size is compared with
  (tests/test_golden_large_gpu.py::test_c4_depth16_parameter_shift_against_the_c_oracle)."""
  n, layers, n_terms = 24, 16, 24
  rng = np.random.default_rng(2416)
  gates, names = O.hea_gates(n, layers, "b")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(1, n)).astype(np.int8)
  op = O.random_pauli_op(n, 512, 24)[:n_terms]
  t0 = time.time()
  term_values = C.expectation(n, gates, params, bits, [[t] for t in op])[0].astype(np.float64)
  print(f"c4d16 C forward: {time.time() - t0:.0f} s; sum {term_values.sum():.6f}", flush=True)
  t0 = time.time()
  vals, grad = C.expectation_vjp(n, gates, params, bits, [op], np.ones((1, 1), np.float32))
  print(f"c4d16 C adjoint: {time.time() - t0:.0f} s; |grad|_inf {np.abs(grad).max():.4f}", flush=True)
  assert abs(float(vals[0, 0]) - term_values.sum()) < 5e-5 * sum(abs(c) for c, _, _ in op)
  save("c4_n24_d16_shift.npz", n=n, layers=layers, gates=np.array(gates, dtype=np.float64), params=params, bits=bits,
       ops=pack_ops([op]), term_values=term_values, grad=grad.astype(np.float64))


if __name__ == "__main__":
  for which in sys.argv[1:] or ["c3", "c4", "c5"]:
    {"c3": make_c3, "c4": make_c4, "c5": make_c5, "c4qmhl": make_c4_qmhl, "c5vjp": make_c5_vjp,
     "c4d16": make_c4_d16}[which]()
