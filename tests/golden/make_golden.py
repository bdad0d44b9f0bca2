"""Generates the golden vectors in this directory with the numpy oracle
(oracle/qhbm_oracle.py, complex128).  Run from the repo root:

    python tests/golden/make_golden.py

The reference itself cannot be imported here (tensorflow / tensorflow-quantum /
cirq are absent), so these vectors are outputs of the
oracle, which is pinned to the reference's closed-form known answers by
tests/test_oracle_kat.py.  Each .npz holds inputs AND expected outputs; the flat
gate lists are stored as float arrays [G, 6] = (kind, q0, q1, param_idx, scalar,
offset) and Pauli ops as [T, 4] = (op_index, coeff, x_mask, z_mask).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import qhbm_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pack_ops(ops):
  rows = []
  for k, op in enumerate(ops):
    for c, x, z in op:
      rows.append((k, c, x, z))
  return np.array(rows, dtype=np.float64)


def save(name, **arrays):
  np.savez_compressed(os.path.join(HERE, name), **arrays)
  print("wrote", name)


def hea_case(n, layers, seed):
  rng = np.random.default_rng(seed)
  gates, names = O.hea_gates(n, layers, "g")
  params = rng.uniform(-1, 1, len(names))          # tests/test_util.py:76-77,88
  bits = O.all_bitstrings(n)
  sum_z = [[O.pauli_term(1.0, [(q, "Z")]) for q in range(n)]]  # qnn_test.py:187-190
  ops = sum_z + [O.tfim_ring_op(n)] + ([O.xxz_chain_op(n)] if n > 1 else [])
  vals, jac = O.expectation_jacobian(n, gates, params, bits, ops)
  # modular Hamiltonians (qnn_test.py:266-369 pattern: bit . U . V^dagger, Z shards)
  v_gates, v_names = O.hea_gates(n, layers, "v")
  p_u = len(names)
  v_shift = [(k, q0, q1, p + p_u, s, o) for (k, q0, q1, p, s, o) in v_gates]
  v_params = rng.uniform(-1, 1, len(v_names))
  total = gates + O.inverse_gates(v_shift)
  all_params = np.concatenate([params, v_params])
  out = dict(n=n, layers=layers, gates=np.array(gates, dtype=np.float64), params=params,
             bits=bits, ops=pack_ops(ops), values=vals, jacobian=jac,
             total_gates=np.array(total, dtype=np.float64), total_params=all_params)
  for tag, shards in (("bernoulli", O.bernoulli_shards(n)), ("kobe2", O.kobe_shards(n, min(2, n)))):
    svals, sjac = O.expectation_jacobian(n, total, all_params, bits, shards)
    out[f"{tag}_shards"] = pack_ops(shards)
    out[f"{tag}_values"] = svals
    out[f"{tag}_jacobian"] = sjac
  return out


def main():
  for n in (2, 3, 4, 6):
    for layers in (1, 2):
      save(f"hea_n{n}_l{layers}.npz", **hea_case(n, layers, 100 * n + layers))
  # (iii) one n = 12 forward in both bit-order modes
  n = 12
  rng = np.random.default_rng(1212)
  gates, names = O.hea_gates(n, 2, "g")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(6, n)).astype(np.int8)
  ops = [[O.pauli_term(float(q + 1), [(q, "Z")]) for q in range(n)], O.tfim_ring_op(n)]
  save("hea_n12_bit_order.npz", n=n, gates=np.array(gates, dtype=np.float64), params=params, bits=bits,
       ops=pack_ops(ops), values_direct=O.expectation(n, gates, params, bits, ops, False),
       values_tfq_compat=O.expectation(n, gates, params, bits, ops, True),
       tfq_permutation=np.array(O.tfq_bit_permutation(n)))
  # (iv) VQT loss and gradients for BASELINE config 1 on a GIVEN multiset (sampler excluded)
  n, layers = 4, 2
  rng = np.random.default_rng(41)
  gates, names = O.hea_gates(n, layers, "c1")
  params = rng.uniform(-1, 1, len(names))
  thetas = rng.uniform(-1, 1, n)
  samples = rng.integers(0, 2, size=(32, n)).astype(np.int8)
  beta = 0.8
  target = O.tfim_ring_op(n)
  log_z = float(np.sum(np.log(2 * np.cosh(thetas))))
  loss, dtheta, dparams = O.vqt_loss_and_grads(
      n, gates, params, samples, target, beta, lambda b: O.bernoulli_energy(b, thetas),
      O.spins_from_bitstrings, log_z)
  save("vqt_c1.npz", n=n, gates=np.array(gates, dtype=np.float64), params=params, thetas=thetas,
       samples=samples, beta=beta, target=pack_ops([target]), log_partition=log_z, loss=loss,
       dtheta=dtheta, dparams=dparams)
  # every gate kind on 5 qubits
  rng = np.random.default_rng(77)
  n, n_params = 5, 6
  gates = []
  for _ in range(48):
    kind = int(rng.integers(12))
    q0 = int(rng.integers(n))
    q1 = -1
    if O.gate_num_qubits(kind) == 2:
      q1 = int(rng.integers(n - 1))
      q1 += q1 >= q0
    pidx = int(rng.integers(n_params)) if rng.random() < 0.8 else -1
    gates.append((kind, q0, q1, pidx, float(rng.uniform(-1.5, 1.5)) if pidx >= 0 else 0.0,
                  float(rng.uniform(-0.5, 0.5))))
  params = rng.uniform(-1, 1, n_params)
  ops = [O.random_pauli_op(n, 8, 5, 0.4), O.xxz_chain_op(n)]
  bits = O.all_bitstrings(n)
  vals, jac = O.expectation_jacobian(n, gates, params, bits, ops)
  save("all_kinds_n5.npz", n=n, gates=np.array(gates, dtype=np.float64), params=params, bits=bits,
       ops=pack_ops(ops), values=vals, jacobian=jac)


if __name__ == "__main__":
  main()
