#!/usr/bin/env python3
"""Benchmark of the expectation hot path (BASELINE.json metric).

One "step" = one VQT step's engine work over one batch of synthetic input:
values [U, 1] of the target Hamiltonian AND the adjoint vector-Jacobian
product (the [P] gradient of the VQT loss w.r.t. the circuit parameters), for
U EBM bitstrings already resident in HBM.  With --gpus N the batch is sharded
over N ranks (one process per GPU) and each step ends with the exchange the
path really has: an RCCL all-reduce of the [P] gradient and an all-gather of
the per-state expectations.

  value = (states over all ranks) x (Pauli terms) / step time      [evals/s]

Workload (config.workload): BASELINE.json configs[2] as the metric states it --
20-qubit XXZ chain, depth-16 hardware-efficient ansatz, a 4096-sample step.
The whole batch fits one MI355X (4096 x 8 MiB x (psi, lambda) = 64 GiB), so
N = 1 runs all 4096 states and N GPUs split the SAME 4096 (strong scaling,
--states-total; N = 8 is exactly the config's 512 states per GPU).
--states-per-gpu S switches to weak scaling.  Synthetic inputs:
phi ~ U[-1,1] seeded, distinct seeded bitstrings (U = B exactly).

Launched without torchrun, `--gpus N` (N > 1) starts its own N ranks through
`python -m torch.distributed.run` as a child process -- decided before anything
touches the GPU -- and fails loudly if fewer than N GPUs are visible.
"""
import argparse
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "qhbm-library_amd")):
  if _p not in sys.path:
    sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# fp32 peak: the vector rate (v_pk_fma_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz), which is also the
# dense f32-input MFMA peak on gfx950 (same guide, "Peak FP32 (vector)" / "(matrix)": 157.3 TFLOP/s)
FP32_PEAK_TFLOPS = 157.3


def hea_gates(n, layers, name="b"):
  """tests/test_util.py:25-67 of the reference; parameters in sorted-symbol
  order (circuit.py:201-204).  Kept here so the product bench does not import
  the oracle for its workload."""
  from qhbmlib_amd import _engine as E
  names = []
  for layer in range(layers):
    for q in range(n):
      names += [f"sx_{name}_{layer}_{q}", f"sz_{name}_{layer}_{q}"]
    for k in range(len(range(0, n - 1, 2))):
      names.append(f"sc_{name}_{layer}_{2 * k}")
    for k in range(len(range(1, n - 1, 2))):
      names.append(f"sc_{name}_{layer}_{2 * k + 1}")
  order = {s: i for i, s in enumerate(sorted(names))}
  gates = []
  for layer in range(layers):
    for q in range(n):
      gates.append((E.GATE_XPOW, q, -1, order[f"sx_{name}_{layer}_{q}"], 1.0, 0.0))
      gates.append((E.GATE_ZPOW, q, -1, order[f"sz_{name}_{layer}_{q}"], 1.0, 0.0))
    for k, q0 in enumerate(range(0, n - 1, 2)):
      gates.append((E.GATE_CZPOW, q0, q0 + 1, order[f"sc_{name}_{layer}_{2 * k}"], 1.0, 0.0))
    for k, q0 in enumerate(range(1, n - 1, 2)):
      gates.append((E.GATE_CZPOW, q0, q0 + 1, order[f"sc_{name}_{layer}_{2 * k + 1}"], 1.0, 0.0))
  return gates, len(names)


def xxz_op(n, delta=0.5):
  terms = []
  for i in range(n - 1):
    m = (1 << i) | (1 << (i + 1))
    terms.append((1.0, m, 0))      # X X
    terms.append((1.0, m, m))      # Y Y
    terms.append((delta, 0, m))    # Z Z
  return terms


def tfim_op(n, bias=1.0):
  terms = [(-bias, 1 << i, 0) for i in range(n)]
  terms += [(-1.0, 0, (1 << i) | (1 << ((i + 1) % n))) for i in range(n)]
  return terms


def random_pauli_op(n, terms, seed, p_identity=0.75):
  """`terms` Pauli strings, each qubit in {I,X,Y,Z} with P(I) = 0.75
  (at least one non-identity factor), coefficients N(0,1)."""
  rng = np.random.default_rng(seed)
  out = []
  while len(out) < terms:
    kinds = rng.choice(4, size=n, p=[p_identity] + [(1 - p_identity) / 3] * 3)
    if not kinds.any():
      continue
    x = sum(1 << q for q in range(n) if kinds[q] in (1, 2))
    z = sum(1 << q for q in range(n) if kinds[q] in (2, 3))
    out.append((float(rng.normal()), x, z))
  return out


def distinct_bitstrings(n, count, seed):
  rng = np.random.default_rng(seed)
  if n <= 40:
    vals = rng.choice(1 << n, size=count, replace=False) if (1 << n) >= count else rng.integers(0, 1 << n, size=count)
  else:
    vals = rng.integers(0, 1 << 40, size=count)
  bits = ((vals[:, None] >> np.arange(n - 1, -1, -1)[None, :]) & 1).astype(np.int8)
  return bits


def cpu_baseline(n, gates, n_params, ops, params, bits, upstream, mode, checker_states=2):
  """The CPU baseline BASELINE.md section 3 describes, timed on this host's cores on a bounded sample of the TIMED batch
  itself (its first K bitstrings, same parameters): oracle/qhbm_cpu_diag.c -- the oracle's fp32 statevector algorithm
  with merged diagonal runs, an AVX2 one-qubit kernel and fused adjoint steps, one state per thread -- a reported
  baseline, never the product (`kind: "port"`, `variant: "port+diag"`).  The CHECKER stays the gate-by-gate restatement oracle/qhbm_cpu.c:
  it runs on the first `checker_states` of those states (threads inside a state), `parity_check` compares the engine
  with IT, and the record carries how far the timed path is from it on the same states.
  Returns the record, the checker's values [k, n_ops] and [P] VJP (upstream 1/k) and k."""
  from oracle import qhbm_cpu as C
  cores = min(C.max_threads(), os.cpu_count() or 1)
  states = bits.shape[0]
  up = np.ascontiguousarray(upstream, np.float32)
  t0 = time.perf_counter()
  C.expectation_vjp_diag(n, gates, params, bits, ops, None if mode == "forward" else up, n_threads=cores)
  dt = time.perf_counter() - t0
  k = max(1, min(checker_states, states))
  up_k = np.ascontiguousarray(up[:k] * (float(states) / float(k)), np.float32)   # upstream theta / k on the checker's rows
  t1 = time.perf_counter()
  if mode == "forward":
    vals, grad = C.expectation(n, gates, params, bits[:k], ops, n_threads=cores), None
    fast_vals, fast_grad = C.expectation_vjp_diag(n, gates, params, bits[:k], ops, None, n_threads=cores)
  else:
    vals, grad = C.expectation_vjp(n, gates, params, bits[:k], ops, up_k, n_threads=cores)
    fast_vals, fast_grad = C.expectation_vjp_diag(n, gates, params, bits[:k], ops, up_k, n_threads=cores)
  dt_check = time.perf_counter() - t1
  return {
      "value": states * sum(len(op) for op in ops) / dt, "unit": "evals/s", "cores": int(min(cores, states)),
      "kind": "port", "variant": "port+diag",   # (kind: the contract's "reference" | "port"; the variant names WHICH port)
      "sample": f"the first {states} states of the timed batch ({mode} step, same parameters), one state "
                f"per thread, {dt:.2f} s wall (oracle/qhbm_cpu_diag.c: merged diagonal runs, AVX2 one-qubit kernel, fused "
                f"adjoint steps; no other gate fusion)",
      "checker": {"what": "oracle/qhbm_cpu.c, gate by gate, threads inside a state", "states": k, "wall_s": dt_check,
                  "max_diff_values_timed_path_vs_checker": float(np.abs(fast_vals - vals).max()),
                  "max_diff_grad_timed_path_vs_checker": (float(np.abs(fast_grad - grad).max()) if grad is not None else None)},
  }, vals, grad, k


def parity_check(eng, E, mode, ops, bits_k, params, timed_vals_k, timed_grad_rows_k, rows_scale, upstream_k, oracle_vals,
                 oracle_grad):
  """The timed workload against the oracle (reference pattern: simulate, compare, assert --
  tests/inference/qnn_test.py:183-264 of the reference).
  Values: rows of the LAST TIMED step.
  Gradient: the per-state gradient rows of the LAST TIMED step's adjoint sweep (qhbm_state_gradients, read after the
  timed region; row u = upstream_u * d<O>_u / d params), its first K rows summed and rescaled from the timed upstream
  1/states_total to the oracle's 1/K (`rows_scale`; the VJP is linear in the upstream), so that the tolerance
  1e-4 * max(1, |grad|_inf) bites -- the timed sweep itself is what is checked.  Parameter-shift steps keep no rows:
  there the engine's VJP of exactly these K states (a second call outside the timed region, same kernels and plans)
  is compared instead, and `grad_from` says so."""
  k = bits_k.shape[0]
  sum_abs = float(max(sum(abs(c) for c, _, _ in op) for op in ops))   # (per observable: the largest bound)
  tol_v = 5e-5 * sum_abs
  err_v = float(np.abs(timed_vals_k - oracle_vals).max())
  out = {"states": k, "max_err_values": err_v, "tol_values": tol_v,
         "values_from": "rows of the last timed step vs oracle/qhbm_cpu.c on the same bitstrings and parameters"}
  ok = err_v <= tol_v
  if oracle_grad is not None:
    if timed_grad_rows_k is not None:
      g = timed_grad_rows_k.astype(np.float64).sum(0) * rows_scale
      grad_from = ("rows of the last timed step (qhbm_state_gradients of the timed adjoint sweep, first K rows summed, "
                   "rescaled from upstream 1/states_total to 1/K) vs the oracle's adjoint VJP with upstream 1/K")
    else:
      up = torch.from_numpy(np.ascontiguousarray(upstream_k, np.float32)).cuda()
      method = E.GRAD_PARAMETER_SHIFT if mode == "shift" else E.GRAD_ADJOINT
      _, g = eng.expectation_vjp(torch.from_numpy(bits_k).cuda(), params, up, method=method)
      g = g.double().cpu().numpy()
      torch.cuda.synchronize()
      grad_from = ("engine VJP of these K states (call outside the timed region, upstream 1/K: this mode keeps no "
                   "per-state rows) vs the oracle's adjoint VJP with the same upstream")
    want = oracle_grad.astype(np.float64)
    gnorm = float(np.abs(want).max())
    tol_g = 1e-4 * max(1.0, gnorm)
    err_g = float(np.abs(g - want).max())
    out.update({"max_err_grad": err_g, "tol_grad": tol_g, "grad_inf_norm": gnorm, "grad_from": grad_from})
    ok = ok and err_g <= tol_g
  out["ok"] = bool(ok)
  return out


def gpus_visible_without_hip():
  """GPU count read from the KFD topology (sysfs), narrowed by HIP_/ROCR_VISIBLE_DEVICES -- no HIP or HSA
  call, so the launching process holds no GPU context.  None when sysfs cannot tell."""
  base = "/sys/class/kfd/kfd/topology/nodes"
  try:
    count = 0
    for node in os.listdir(base):
      with open(os.path.join(base, node, "properties")) as f:
        props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
      if int(props.get("simd_count", "0")) > 0:
        count += 1
  except (OSError, ValueError):
    return None
  for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
    if os.environ.get(var) is not None:
      count = min(count, len([x for x in os.environ[var].split(",") if x.strip()]))
  return count


def self_launch(args):
  """`bench.py --gpus N` outside torchrun: run the N ranks as a child torch.distributed.run (never an
  exec) and exit with its code.  This parent makes no HIP call: the early refusal below counts GPUs
  through sysfs, and every rank checks again with its own runtime (main)."""
  visible = gpus_visible_without_hip()
  if visible is not None and visible < args.gpus and os.environ.get("QHBM_BENCH_SHARE_DEVICE") != "1":
    raise SystemExit(f"bench.py --gpus {args.gpus}: only {visible} GPU(s) visible -- refusing to report a "
                     f"{args.gpus}-GPU number from fewer devices")
  with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  raise SystemExit(subprocess.call(cmd))


def kernel_sources_sha16():
  """sha256 of the engine's sources (csrc/*.hip, *.inc, *.cpp, *.h, in name order) WITHOUT their comments and white space:
  stored with every profile, so that a bench line can say when its stored counters were taken on OTHER code than the code
  it ran (an edited comment does not count)."""
  import re  # pylint: disable=import-outside-toplevel
  csrc = os.path.join(ROOT, "qhbm-library_amd", "csrc")
  h = hashlib.sha256()
  try:
    for name in sorted(os.listdir(csrc)):
      if name.endswith((".hip", ".inc", ".cpp", ".h")) and not name.startswith("_"):
        with open(os.path.join(csrc, name), "r", encoding="utf-8", errors="replace") as f:
          text = f.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)   # block comments
        text = re.sub(r"//[^\n]*", " ", text)                 # line comments (no source string holds "//")
        h.update(name.encode() + b"\0" + "".join(text.split()).encode())
  except OSError:
    return None
  return h.hexdigest()[:16]


def stored_profile(name, n, layers, hamiltonian, mode):
  """profiles/<name>.json if it was taken on this workload (same circuit, observable, mode)."""
  path = os.path.join(ROOT, "profiles", name)
  try:
    with open(path) as f:
      tj = json.load(f)
  except (OSError, ValueError):
    return None
  same = (tj.get("n_qubits") == n and tj.get("layers", 16) == layers and
          tj.get("hamiltonian", "xxz") == hamiltonian and tj.get("mode", "vqt") == mode)
  return tj if same else None


def mirror_model(n, layers, samples, ham_kind, ebm_kind):
  """The QHBM of a BASELINE config through the host mirror (HEA circuit of tests/test_util.py:25-67, Bernoulli or KOBE-2
  EBM on the device, TFIM ring or XXZ chain) and ONE drawn sample multiset: (circuit, energy, e_inference, qhbm,
  hamiltonian, variables, unique rows, counts)."""
  from qhbmlib_amd import inference, ir, models, utils  # pylint: disable=import-outside-toplevel
  qubits = ir.GridQubit.rect(1, n)
  pqc = ir.Circuit()
  for layer in range(layers):                      # tests/test_util.py:25-67 of the reference
    for q, qubit in enumerate(qubits):
      pqc += [ir.X(qubit)**ir.Symbol(f"sx_b_{layer}_{q}"), ir.Z(qubit)**ir.Symbol(f"sz_b_{layer}_{q}")]
    for k, (q0, q1) in enumerate(zip(qubits[::2], qubits[1::2])):
      pqc += ir.CZPowGate(ir.Symbol(f"sc_b_{layer}_{2 * k}"))(q0, q1)
    for k, (q0, q1) in enumerate(zip(qubits[1::2], qubits[2::2])):
      pqc += ir.CZPowGate(ir.Symbol(f"sc_b_{layer}_{2 * k + 1}"))(q0, q1)
  torch.manual_seed(1234)
  circuit = models.DirectQuantumCircuit(pqc, tfq_compat_bit_order=False).to("cuda")
  with torch.no_grad():
    circuit.trainable_variables[0].uniform_(-1, 1)
  energy = (models.BernoulliEnergy(list(range(n))) if ebm_kind == "bernoulli" else models.KOBE(list(range(n)), 2)).to("cuda")
  with torch.no_grad():
    energy.post_process[0].kernel.uniform_(-0.1, 0.1)   # high entropy: U close to the sample count
  e_inf = (inference.BernoulliEnergyInference if ebm_kind == "bernoulli" else inference.AnalyticEnergyInference)(
      energy, samples, initial_seed=7)
  qhbm = inference.QHBM(e_inf, inference.AnalyticQuantumInference(circuit))
  ham = ir.PauliSum()
  if ham_kind == "tfim":
    for i, q in enumerate(qubits):
      ham += -1.0 * ir.PX(q)
      ham += -1.0 * ir.PZ(q) * ir.PZ(qubits[(i + 1) % n])
  else:
    for a, b in zip(qubits, qubits[1:]):
      ham += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
  variables = list(energy.parameters()) + circuit.trainable_variables
  with torch.no_grad():
    drawn = e_inf.sample(samples).cuda()
  rows, _, counts = utils.unique_bitstrings_with_counts(drawn)   # the fixed multiset of every timed step
  return circuit, energy, e_inf, qhbm, ham, variables, rows, counts


def mirror_step_sample(n, layers, samples, ham_kind, steps=3):
  """`vqt_step_through_mirror` of the DEFAULT line (round 5's review, "what's missing" 4): BASELINE.md section 3 defines
  "VQT step time" as loss + both gradients through `vqt()`, sampler excluded -- the engine's share is `ms_per_step`, this is
  the whole: `steps` eager `inference.vqt(qhbm, [H], 1.0)` + `backward()` on one fixed multiset of `samples` samples of a
  KOBE-2 (Bernoulli below 14 qubits) EBM, after one untimed step.  Runs AFTER the timed region, the probe and the parity
  check, on an engine of its own (the bench's is closed first)."""
  from qhbmlib_amd import inference  # pylint: disable=import-outside-toplevel
  _, _, e_inf, qhbm, ham, variables, rows, counts = mirror_model(n, layers, samples, ham_kind, "kobe2" if n >= 14 else "bernoulli")

  def step():
    for v in variables:
      v.grad = None
    with e_inf.fixed_samples(rows, counts):
      loss = inference.vqt(qhbm, [ham], 1.0)
      loss.backward()
    return loss

  step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    loss = step()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / steps * 1e3
  return {"ms_per_step": ms, "steps": steps, "samples": samples, "unique_bitstrings": int(rows.shape[0]),
          "loss": float(loss.detach()), "what": "inference.vqt(qhbm, [H], 1.0) + backward() through the host mirror, eager, fixed "
          "sample multiset (sampler excluded), EBM on the device; `python bench.py --through-mirror c3` is the full measurement"}


def mirror_bench(args):
  """`--through-mirror`: one JSON line with `mirror_step_ms` (eager), `captured_step_ms` (hipGraph replay) and the
  engine's `engine_ms_per_step` on the same unique rows.  Reference of the step: vqt_loss.py:25-55, ebm.py:262-329."""
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
  from qhbmlib_amd import inference  # pylint: disable=import-outside-toplevel
  torch.cuda.set_device(0)
  cfg = {"c1": dict(n=4, layers=2, samples=32, ham="tfim", ebm="bernoulli", label="BASELINE configs[0]"),
         "c2": dict(n=12, layers=8, samples=1024, ham="tfim", ebm="bernoulli", label="BASELINE configs[1]"),
         "c3": dict(n=20, layers=16, samples=4096, ham="xxz", ebm="kobe2", label="BASELINE configs[2]")}[args.through_mirror]
  n, layers, samples = cfg["n"], cfg["layers"], cfg["samples"]
  circuit, energy, e_inf, qhbm, ham, variables, rows, counts = mirror_model(n, layers, samples, cfg["ham"], cfg["ebm"])
  n_unique = int(rows.shape[0])

  def eager_step():
    for v in variables:
      v.grad = None
    with e_inf.fixed_samples(rows, counts):
      loss = inference.vqt(qhbm, [ham], 1.0)
      loss.backward()
    return loss.detach()

  def timed(fn):
    for _ in range(args.warmup):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3, out

  mirror_ms, loss_eager = timed(eager_step)
  grads_eager = [v.grad.detach().clone() for v in variables]
  # the engine's own time on the same rows: values + adjoint VJP (the default line's step)
  (eng,) = list(qhbm.q_inference._engines._engines.values())   # pylint: disable=protected-access
  upstream = (counts.float() / counts.sum()).reshape(-1, 1).contiguous()
  phi = circuit.symbol_values.detach()
  engine_ms, _ = timed(lambda: eng.expectation_vjp(rows, phi, upstream))
  # hipGraph replay of the padded step
  step = inference.CapturedLoss(lambda: inference.vqt(qhbm, [ham], 1.0), [e_inf], variables)
  loss_padded_eager = step.eager([(rows, counts)]).clone()
  grads_padded_eager = [v.grad.detach().clone() for v in variables]
  captured_ms, loss_replay = timed(lambda: step([(rows, counts)]))
  torch.cuda.synchronize()
  same_bits = bool(torch.equal(loss_replay, loss_padded_eager)) and all(
      torch.equal(v.grad, g) for v, g in zip(variables, grads_padded_eager))
  replay_vs_padded = [float((loss_replay - loss_padded_eager).abs())] + [
      float((v.grad - g).abs().max()) for v, g in zip(variables, grads_padded_eager)]
  drift = max([float((loss_replay - loss_eager).abs())] +
              [float((v.grad - g).abs().max()) for v, g in zip(variables, grads_eager)])
  replay_only_ms, _ = timed(step.replay)
  with open(os.path.abspath(__file__), "rb") as f:
    bench_sha = hashlib.sha256(f.read()).hexdigest()[:16]
  line = {
      "metric": "VQT step time through the host mirror: inference.vqt(qhbm, H, beta) + backward(), sampler excluded",
      "value": mirror_ms, "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": mirror_ms, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
      "data": "synthetic",
      "config": {"workload": (f"{cfg['label']} through the host mirror: {n}-qubit {cfg['ham'].upper()}, {cfg['ebm']} EBM, HEA depth "
                              f"{layers}, {samples} samples = {n_unique} unique bitstrings (fixed multiset, sampler excluded)"),
                 "n_qubits": n, "layers": layers, "samples": samples, "unique_bitstrings": n_unique,
                 "bench_py_sha16": bench_sha, "kernel_sources_sha16": kernel_sources_sha16()},
      "mirror_step_ms": mirror_ms,                 # eager: vqt() + backward() on the unique rows
      "captured_step_ms": captured_ms,             # CapturedLoss: buffers refilled + hipGraph replay, padded to `samples` rows
      "captured_replay_only_ms": replay_only_ms,   # the graph replay alone
      "engine_ms_per_step": engine_ms,             # qhbm_expectation_vjp on the same unique rows
      "mirror_over_engine": mirror_ms / engine_ms, "captured_over_engine": captured_ms / engine_ms,
      "replay_equals_padded_eager_bitwise": same_bits,
      "replay_vs_padded_eager_abs_diffs": replay_vs_padded,   # [loss, grad of every variable]
      "max_abs_diff_replay_vs_unpadded_eager": drift,
  }
  print(json.dumps(line), flush=True)
  if not same_bits or not drift < 1e-4:
    raise SystemExit(3)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--qubits", type=int, default=20)
  ap.add_argument("--layers", type=int, default=16)
  ap.add_argument("--states-total", type=int, default=4096,
                  help="states of one step over ALL ranks (strong scaling; BASELINE config 3: 4096)")
  ap.add_argument("--states-per-gpu", type=int, default=0,
                  help="fixed states per rank instead (weak scaling)")
  ap.add_argument("--hamiltonian", choices=["xxz", "xxz3", "tfim", "random512"], default="xxz",
                  help="xxz3: the XXZ chain as THREE observables (its XX, YY and ZZ sums) in one call -- several operators "
                       "per call is the reference's normal usage (tests/inference/qnn_test.py:187-190)")
  ap.add_argument("--mode", choices=["vqt", "forward", "shift", "qmhl"], default="vqt",
                  help="vqt: values + adjoint VJP of one Hamiltonian; forward: values only; shift: values + "
                       "parameter-shift VJP; qmhl: the engine work of a QMHL step (qmhl_loss.py:33-34) -- the circuit "
                       "U_data then U_model^dagger, the KOBE-2 Z-string shards of the model's modular Hamiltonian as "
                       "observables, the adjoint VJP of sum_k theta_k <Z_k> with respect to the MODEL's parameters only "
                       "(the data circuit is frozen: qhbm_set_gradient_mask)")
  ap.add_argument("--tile-qubits", type=int, default=0)
  ap.add_argument("--adjoint-tile-qubits", type=int, default=0)
  ap.add_argument("--engine-option", action="append", default=[], metavar="NAME=VALUE",
                  help="qhbm_set_option knob for experiments (repeatable), e.g. adjoint_exchange=0")
  ap.add_argument("--reduction", choices=["allreduce", "ordered"], default="allreduce",
                  help="N > 1 gradient exchange: all-reduce of the [P] gradient (engine level), or the host "
                       "mirror's default -- all-gather of the per-state rows [U, P], added in global state order "
                       "(bit-identical for any N; AnalyticQuantumInference(ordered_reduction=True))")
  ap.add_argument("--balance", choices=["equal", "measured"], default="equal",
                  help="N > 1, strong scaling: equal row blocks (default), or blocks proportional to every rank's MEASURED "
                       "speed -- one probing step on the equal blocks, kernel time per state all-gathered once "
                       "(parallel.measured_weights) -- so that a GPU that sustains a lower clock gets fewer states")
  ap.add_argument("--cpu-sample-states", type=int, default=64)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--verify", dest="verify", action="store_true", default=None,
                  help="rank 0 re-evaluates the whole batch alone (outside the timed region) and compares with the "
                       "sharded result; default: on for --gpus N > 1, off for N = 1")
  ap.add_argument("--no-verify", dest="verify", action="store_false")
  ap.add_argument("--no-mirror-step", action="store_true",
                  help="skip `vqt_step_through_mirror` (three eager vqt() + backward() steps through the host mirror after "
                       "the timed region: BASELINE.md's 'VQT step time', sampler excluded)")
  ap.add_argument("--through-mirror", choices=["c1", "c2", "c3"], default=None,
                  help="a SECOND, separately labelled measurement (the default line is unchanged): the step a user calls -- "
                       "inference.vqt(qhbm, H, beta) + backward() through the host mirror on a fixed sample multiset (sampler "
                       "excluded: BASELINE.md section 3 'VQT step time') -- for BASELINE configs[0], [1] or [2], eager and as a "
                       "replayed hipGraph (inference.CapturedLoss), beside the engine's own time on the same rows")
  args = ap.parse_args()
  if args.through_mirror:
    if args.gpus != 1:
      raise SystemExit("bench.py --through-mirror is a one-GPU measurement")
    return mirror_bench(args)

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    self_launch(args)
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    if rank == 0:
      print(f"bench.py: --gpus {args.gpus} but launched with WORLD_SIZE={world}; running {world} ranks",
            file=sys.stderr)
    args.gpus = world
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
  # Test hooks (tests/test_bench_gpu.py runs two ranks on ONE GPU): QHBM_BENCH_SHARE_DEVICE=1 puts
  # every rank on cuda:0 and QHBM_BENCH_BACKEND=gloo carries the collectives through the host.
  if os.environ.get("QHBM_BENCH_SHARE_DEVICE") == "1":
    local_rank = 0
  elif torch.cuda.device_count() < world:
    raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPU(s) visible -- refusing to "
                     f"report a {world}-GPU number from fewer devices")
  backend = os.environ.get("QHBM_BENCH_BACKEND", "nccl")
  torch.cuda.set_device(local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
      # bind the communicator to this rank's GPU up front (eager RCCL init on the right device, no guessing
      # from the first collective's tensors)
      dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)

  from qhbmlib_amd import _engine as E  # pylint: disable=import-outside-toplevel
  from qhbmlib_amd import parallel  # pylint: disable=import-outside-toplevel

  n, layers = args.qubits, args.layers
  weak = args.states_per_gpu > 0
  total_states = args.states_per_gpu * world if weak else args.states_total
  blocks = parallel.partition(total_states, world)
  lo, hi = blocks[rank]
  spg = hi - lo
  rng = np.random.default_rng(1234)
  grad_mask = None
  if args.mode == "qmhl":
    # qmhl(data, qhbm) = <K_model>_data + log Z (qmhl_loss.py:33-34); data.expectation(K_model) runs the total circuit
    # bit-inject . U_data(phi_d) . U_model(phi_m)^dagger and measures the Z-string shards of the model's energy
    # (qnn.py:120-139, energy.py:200-209: all 1- and 2-subsets in itertools.combinations order), combined by theta.
    import itertools  # pylint: disable=import-outside-toplevel
    g_data, p_data = hea_gates(n, layers, "d")
    g_model, p_model = hea_gates(n, layers, "m")
    # inverse circuit = reversed gate list, scalar and offset negated, same variables (circuit.py:164-176)
    inverse = [(k, q0, q1, p_data + pi, -sc, -off) for (k, q0, q1, pi, sc, off) in reversed(g_model)]
    gates, n_params = g_data + inverse, p_data + p_model
    ops = [[(1.0, 0, sum(1 << q for q in subset))] for order in (1, 2) for subset in itertools.combinations(range(n), order)]
    thetas = rng.uniform(-1, 1, len(ops)).astype(np.float32)
    grad_mask = np.arange(n_params) >= p_data
    ham_name, args.hamiltonian = f"KOBE-2 shards ({len(ops)} Z strings)", "kobe2_shards"
  else:
    gates, n_params = hea_gates(n, layers)
    if args.hamiltonian == "xxz3":
      whole = xxz_op(n)
      ops = [whole[0::3], whole[1::3], whole[2::3]]          # the XX, the YY and the ZZ terms
    else:
      ops = [{"xxz": xxz_op, "tfim": tfim_op, "random512": lambda m: random_pauli_op(m, 512, 24)}[args.hamiltonian](n)]
    thetas = np.ones(len(ops), np.float32)
    ham_name = {"xxz": "XXZ(delta=0.5) open chain", "tfim": "TFIM ring", "xxz3": "XXZ chain as 3 observables (XX, YY, ZZ sums)",
                "random512": "random 512-term Pauli sum"}[args.hamiltonian]
  n_ops = len(ops)
  params_np = rng.uniform(-1, 1, n_params).astype(np.float32)
  all_bits = distinct_bitstrings(n, total_states, 4321)
  bits = torch.from_numpy(all_bits[lo:hi]).cuda()
  params = torch.from_numpy(params_np).cuda()
  # upstream[u, k] = theta_k / states: the weight of <O_k>_u in the loss (one Hamiltonian: 1 / states)
  upstream = (torch.from_numpy(thetas).cuda() / float(total_states)).repeat(spg, 1).contiguous()

  eng = E.Engine(local_rank)
  if args.tile_qubits:
    eng.set_option("tile_qubits", args.tile_qubits)
  if args.adjoint_tile_qubits:
    eng.set_option("adjoint_tile_qubits", args.adjoint_tile_qubits)
  for item in args.engine_option:
    key, _, val = item.partition("=")
    eng.set_option(key, int(val))
  eng.set_circuit(n, gates, n_params)
  eng.set_observables(ops)
  if grad_mask is not None:
    eng.set_gradient_mask(grad_mask)
  eng.set_option("profile_events", 1)
  fwd_passes, bwd_passes = eng.num_passes()

  def step():
    if args.mode == "forward":
      vals = eng.expectation(bits, params)
      grad = None
    elif args.mode == "qmhl":
      vals, grad = eng.expectation_vjp(bits, params, upstream)
    else:
      vals, grad = eng.expectation_vjp(bits, params, upstream,
                                       method=E.GRAD_PARAMETER_SHIFT if args.mode == "shift" else E.GRAD_ADJOINT)
    if world > 1:
      if grad is not None and args.reduction == "ordered" and args.mode in ("vqt", "qmhl"):
        rows = eng.state_gradients(spg) if spg else grad.new_zeros((0, grad.numel()))
        rows = parallel.all_gather_rows(rows, blocks)
        grad = rows.to(torch.float64).sum(0).to(torch.float32)
      elif grad is not None:
        parallel.all_reduce_sum(grad)
      vals = parallel.all_gather_rows(vals, blocks)
    return vals, grad

  shard_weights = None
  if args.balance == "measured" and world > 1 and not weak:
    # one probing step on the equal blocks (after one that builds plans and workspaces): this rank's kernel time per state
    step()
    torch.cuda.synchronize()
    eng.kernel_time_ms(reset=True)
    step()
    torch.cuda.synchronize()
    probe_kt = eng.kernel_time_ms(reset=True)
    per_state = (probe_kt["fwd_ms"] + probe_kt["bwd_ms"] + probe_kt["obs_ms"]) * 1e-3 / max(1, spg)
    shard_weights = parallel.measured_weights(per_state if spg > 0 and per_state > 0 else 1.0)
    blocks = parallel.partition(total_states, world, shard_weights)
    lo, hi = blocks[rank]
    spg = hi - lo
    bits = torch.from_numpy(all_bits[lo:hi]).cuda()
    upstream = (torch.from_numpy(thetas).cuda() / float(total_states)).repeat(spg, 1).contiguous()
  for _ in range(args.warmup):
    step()
  torch.cuda.synchronize()
  eng.kernel_time_ms(reset=True)
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    vals, grad = step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if world > 1:
    tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
  kt = eng.kernel_time_ms(reset=True)
  # the chip's sustained packed-fp32 rate and clock, probed straight after the timed region (as warm as the timed
  # kernels left it): the ATTAINABLE compute ceiling of this run, next to the nominal 157.3 TFLOP/s
  probe = None
  if rank == 0:
    try:
      probe = eng.clock_probe()
    except Exception as exc:  # pylint: disable=broad-except
      probe = {"error": f"{type(exc).__name__}: {exc}"}
  if args.verify is None:
    args.verify = world > 1
  # per-state gradient rows of the LAST TIMED step, read before anything else runs on the engine (parity_check)
  timed_grad_rows = None
  if rank == 0 and args.mode in ("vqt", "qmhl") and not args.no_cpu_baseline and spg > 0:
    k_rows = max(1, min(args.cpu_sample_states, spg, os.cpu_count() or 1))
    timed_grad_rows = eng.state_gradients(spg)[:k_rows].float().cpu().numpy()

  # which physical device each rank ran on (PCI bus id), as the ranks themselves report it
  prop = torch.cuda.get_device_properties(local_rank)
  my_dev = f"{local_rank}:{getattr(prop, 'pci_bus_id', 0):02x}:{getattr(prop, 'pci_device_id', 0):02x}"
  device_ids = [my_dev]
  if world > 1:
    device_ids = [None] * world
    dist.all_gather_object(device_ids, my_dev)

  # every rank's own kernel time per step (HIP events on its launch stream) and block size: a straggler shows up here
  my_kernel_ms = (kt["fwd_ms"] + kt["bwd_ms"] + kt["obs_ms"]) / max(1, args.steps)
  per_rank = [(my_kernel_ms, spg)]
  if world > 1:
    per_rank = [None] * world
    dist.all_gather_object(per_rank, (my_kernel_ms, spg))

  parity_failed = False
  if rank == 0:
    n_terms = sum(len(op) for op in ops)
    evals_per_step = total_states * n_terms
    ms_per_step = dt / args.steps * 1e3
    # ---- roofline of the dominant kernel -------------------------------------------------------
    # Bytes a launch must move: every tile it touches read once and written once (qhbm_traffic_model;
    # it agrees with the rocprofv3 FETCH_SIZE / WRITE_SIZE counters of profiles/ to < 1 %).  The
    # per-gate byte model of SURVEY.md 8(d) -- what an UNFUSED gate-by-gate sweep would move -- is
    # reported beside it as `unfused_bytes_per_launch`; their ratio is the fusion factor, not a
    # roofline fraction.
    n_gate = len(gates)
    amp = float(1 << n)
    shift_factor = (1 + 2 * n_params) if args.mode == "shift" else 1
    fwd_unfused = spg * (16.0 * n_gate + 8.0 * n_terms + 8.0) * amp * shift_factor
    bwd_unfused = spg * 48.0 * n_gate * amp
    adjoint_mode = args.mode in ("vqt", "qmhl")
    tm = eng.traffic_model(spg, with_vjp=adjoint_mode)
    fm = eng.flop_model(spg, with_vjp=adjoint_mode)
    # the dominant kernel family of the step: forward passes, adjoint passes, or the observable kernel (lambda = O psi /
    # values: config 4's 480 X-masks make it the largest)
    families = {
        "pass_fwd_kernel": (kt["fwd_launches"], kt["fwd_ms"], fwd_unfused, tm["fwd_bytes"] * shift_factor, fm["fwd_flops"] * shift_factor),
        "pass_adj_kernel": (kt["bwd_launches"], kt["bwd_ms"], bwd_unfused, tm["bwd_bytes"], fm["bwd_flops"]),
        "apply_observable_kernel": (kt["obs_launches"], kt["obs_ms"], spg * 8.0 * n_terms * amp * shift_factor,
                                    tm["obs_bytes"] * shift_factor, fm["obs_flops"] * shift_factor),
    }
    name = max(families, key=lambda k: families[k][1])
    launches, ms, unfused, model, flops = families[name]
    kernel_label = name
    observable_family = name == "apply_observable_kernel"
    if observable_family:  # which of the two observable kernels ran (qhbm_describe_schedule's last line)
      m = re.search(r"observable kernel: lambda = (\w+) values = ([\w ]+)", eng.describe_schedule())
      if m:
        kernel_label = m.group(1) if adjoint_mode else m.group(2).strip()
    per_step_launches = max(1, launches // max(1, args.steps))
    avg_ms = ms / max(1, launches)
    bytes_per_launch = model / per_step_launches
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # compute roofline of the same kernel: the fp32 operations its gate arithmetic executes
    # (qhbm_flop_model: counted from the plan's instance records, FMA = 2; reductions and address
    # arithmetic not counted) over the same HIP-event launch time, against the fp32 vector peak
    flops_per_launch = flops / per_step_launches
    achieved_tfs = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    hbm_frac, compute_frac = achieved / HBM_PEAK_GBPS, achieved_tfs / FP32_PEAK_TFLOPS
    compute_bound = compute_frac > hbm_frac
    # PMC traffic and VALU utilisation come from committed rocprofv3 runs of THIS command
    # (scripts/profile_bench.sh -> profiles/): stored values, not measured in this run.
    traffic, traffic_src, valu = None, None, None
    def family(profile):  # "pass_adj_kernel" also matches the exchange-layout kernel "pass_adjx_kernel"
      keys = [k for k in (profile or {}) if k.startswith(name[:-len("_kernel")]) and isinstance(profile[k], dict)]
      return max(keys, key=lambda k: profile[k].get("calls", 0)) if keys else None

    tj = stored_profile("traffic.json", n, layers, args.hamiltonian, args.mode)
    if family(tj):
      scale = spg / float(tj["states_per_gpu"])
      traffic = tj[family(tj)]["hbm_bytes_per_launch"] * scale
      traffic_src = (f"stored profile profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                     f"{tj['states_per_gpu']} states, git {tj.get('git_head', '?')}, bench.py sha "
                     f"{tj.get('bench_py_sha16', '?')}), scaled x{scale:g} in states; NOT measured in this run")
    vj = stored_profile("valu.json", n, layers, args.hamiltonian, args.mode)
    if family(vj):
      valu = dict(vj[family(vj)], source=f"stored profile profiles/valu.json (git {vj.get('git_head', '?')})")
    #  the stored counters go stale with every kernel change: say so in the line when they were taken on
    # other engine sources than the ones in this tree
    src_sha = kernel_sources_sha16()
    stale = [nm for nm, pj in (("traffic.json", tj), ("valu.json", vj))
             if family(pj) and pj.get("kernel_sources_sha16") not in (None, src_sha)]
    with open(os.path.abspath(__file__), "rb") as f:
      bench_sha = hashlib.sha256(f.read()).hexdigest()[:16]
    mode_name = {"vqt": "VQT step = values + adjoint VJP", "forward": "forward values only",
                 "shift": "values + parameter-shift VJP",
                 "qmhl": "QMHL step = shard values of U_data U_model^dagger + adjoint VJP w.r.t. the model's parameters"}[args.mode]
    is_c3 = (n, layers, args.hamiltonian, args.mode) == (20, 16, "xxz", "vqt")
    shape = (n, layers, args.hamiltonian)
    if is_c3:
      label = ("BASELINE configs[2] (20-qubit XXZ, depth 16, 4096-sample VQT step)" if total_states == 4096
               else "BASELINE configs[2] circuit at a different batch")
    elif shape == (12, 8, "tfim"):
      label = "BASELINE configs[1] (12-qubit TFIM, depth-8 HEA" + (", 1024 samples)" if total_states == 1024 else ") at another batch")
    elif shape == (24, 16, "random512"):
      label = ("BASELINE configs[3]'s shape (24-qubit random 512-term Pauli sum; depth 16 assumed, SURVEY 8d) with "
               + ("parameter-shift gradients" if args.mode == "shift" else "adjoint gradients"))
    elif shape == (28, 32, "tfim"):
      label = "BASELINE configs[4]'s shape (28-qubit TFIM, depth 32)"
    elif args.mode == "qmhl" and (n, layers) == (20, 16):
      label = "QMHL step at BASELINE configs[2]'s size (20 qubits, depth 16, KOBE-2 model)"
    else:
      label = "custom workload"
    line = {
        "metric": "circuit-expectation evals/sec (samples×Pauli terms) at n qubits; VQT step time",
        "value": evals_per_step / (dt / args.steps),
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak" if weak else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{label}: {n}-qubit {ham_name}, HEA depth {layers} ({n_params} params), "
                         f"{total_states} states in total = {spg} on rank 0 of {world}, {mode_name}"),
            "n_qubits": n, "layers": layers, "states_total": total_states, "states_per_gpu": spg,
            "pauli_terms": n_terms, "observables": n_ops, "hamiltonian": args.hamiltonian,
            "mode": args.mode, "parallelism": f"batch-sharded x{world}",
            # what the collective backend itself reports: a SCALE record shows that RCCL saw N ranks
            "backend": dist.get_backend() if world > 1 else None,
            "backend_world_size": dist.get_world_size() if world > 1 else 1,
            "devices": sorted(set(device_ids)),
            "reduction": (args.reduction if args.mode in ("vqt", "qmhl") else "allreduce") if world > 1 else None,
            # bytes one step's exchange moves per rank (N > 1): the all-gather of the values [U, 1] plus either the
            # all-reduce of the [P] gradient or, with --reduction ordered, the all-gather of the per-state rows [U, P]
            "exchange_bytes": None if world == 1 else int(
                4 * total_states * n_ops + (0 if args.mode == "forward" else
                                            4 * (total_states * n_params
                                                 if (args.reduction == "ordered" and args.mode in ("vqt", "qmhl")) else n_params))),
            "forward_passes": fwd_passes, "adjoint_passes": bwd_passes,
            "bench_py_sha16": bench_sha, "kernel_sources_sha16": kernel_sources_sha16(), "engine_options": args.engine_option,
        },
        "vqt_step_ms": ms_per_step if args.mode == "vqt" else None,
        "qmhl_step_ms": ms_per_step if args.mode == "qmhl" else None,
        "kernel_ms_per_step": {"forward": kt["fwd_ms"] / args.steps, "adjoint": kt["bwd_ms"] / args.steps,
                               "apply_observable": kt["obs_ms"] / args.steps},
        # per rank: kernel time per step and states held -- ranks run in lock step, the slowest one sets the pace
        "per_rank": {"kernel_ms_per_step": [float(k) for k, _ in per_rank], "states": [int(b) for _, b in per_rank],
                     "min": float(min(k for k, _ in per_rank)), "max": float(max(k for k, _ in per_rank)),
                     "argmax": int(max(range(len(per_rank)), key=lambda r: per_rank[r][0])),
                     "balance": args.balance if world > 1 else None,
                     "weights": shard_weights},
        "roofline": {
            # the bound is whichever ceiling the dominant kernel sits closer to; achieved / peak / unit /
            # frac are those of that ceiling, and both are spelled out in "hbm" and "compute"
            "bound": "fp32_valu" if compute_bound else "hbm", "kernel": kernel_label,
            "achieved": achieved_tfs if compute_bound else achieved,
            "peak": FP32_PEAK_TFLOPS if compute_bound else HBM_PEAK_GBPS,
            "unit": "TFLOP/s" if compute_bound else "GB/s",
            "frac": compute_frac if compute_bound else hbm_frac,
            "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_frac},
            "compute": {"flops_per_launch": flops_per_launch, "achieved_TFs": achieved_tfs, "peak": FP32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": compute_frac,
                        # measured in THIS run (qhbm_clock_probe): every SIMD issuing v_pk_fma_f32 from four waves --
                        # what the chip sustains under packed fp32 at the clock it holds right after the timed region
                        "attainable_peak": (probe or {}).get("tflops"),
                        "attainable_frac": (achieved_tfs / probe["tflops"]) if probe and probe.get("tflops") else None,
                        "probe": probe,
                        "flops_definition": "fp32 operations of the gate arithmetic (FMA = 2) from the plan's instance "
                                            "records (qhbm_flop_model), skipped tiles and dead waves excluded; peak = "
                                            "fp32 vector rate = dense f32 MFMA rate on gfx950",
                        "flops_per_step": {"forward": fm["fwd_flops"] * shift_factor, "apply_observable": fm["obs_flops"],
                                           "adjoint": fm["bwd_flops"]},
                        "achieved_TFs_by_kernel": {
                            k: (f / (kt[t] / args.steps * 1e-3) / 1e12 if kt[t] > 0 else None)
                            for k, f, t in (("forward", fm["fwd_flops"] * shift_factor, "fwd_ms"),
                                            ("apply_observable", fm["obs_flops"], "obs_ms"),
                                            ("adjoint", fm["bwd_flops"], "bwd_ms"))}},
            "traffic": traffic, "traffic_source": traffic_src,
            **({"stored_profile_warning": "STALE: " + ", ".join(stale) + " taken on other engine sources than this tree's "
                                          f"(kernel_sources_sha16 {src_sha}): re-run scripts/profile_bench.sh"} if stale else {}),
            "avg_launch_ms": avg_ms, "launches_per_step": per_step_launches,
            "bytes_per_launch": bytes_per_launch,
            "bytes_definition": (
                "the final states read once (+ lambda written once in a VJP call): the least this kernel could move "
                "(qhbm_traffic_model); what it does move through L2 and the fabric is in the committed counter profiles"
                if observable_family else
                "every tile the launch touches read once + written once (qhbm_traffic_model); "
                "achieved = bytes_per_launch / avg_launch_ms (HIP events on the launch stream)"),
            "unfused_bytes_per_launch": unfused / per_step_launches,
            "fusion_factor": unfused / model if model > 0 else None,
            "gates_per_launch": None if observable_family else n_gate / per_step_launches,
            "valu": valu,
            "note": (("the observable kernel is the largest launch of this step: compute.frac counts one complex "
                      "multiply-add per Pauli term and amplitude against the fp32 peak; the block-grouped kernel is bound "
                      "by its LDS pipeline (store burst of the partner block, the masks' reads, barrier skew: DESIGN.md "
                      "section 5.2), the gather kernel by the fabric reads of its partner runs -- neither by HBM nor by VALU issue")
                     if observable_family else
                     ("the pass kernels are bound by fp32 VALU issue (valu.valu_active_frac), not by HBM: "
                      "hbm.frac is the HBM rate of a launch that applies gates_per_launch gates per tile round "
                      "trip, and it FALLS when the scheduler fuses more gates into a launch while the step "
                      "gets faster; compute.frac is the share of the fp32 peak the gate arithmetic reaches "
                      "(DESIGN.md section 5)")),
        },
    }
    if args.verify:
      # the sharded step against one process evaluating every state (same engine, same inputs)
      full_bits = torch.from_numpy(all_bits).cuda()
      full_up = (torch.from_numpy(thetas).cuda() / float(total_states)).repeat(total_states, 1).contiguous()
      if args.mode == "forward":
        ref_vals, ref_grad = eng.expectation(full_bits, params), None
      else:
        ref_vals, ref_grad = eng.expectation_vjp(
            full_bits, params, full_up,
            method=E.GRAD_PARAMETER_SHIFT if args.mode == "shift" else E.GRAD_ADJOINT)
      err_v = float((vals - ref_vals).abs().max())
      err_g = float((grad - ref_grad).abs().max()) if ref_grad is not None else 0.0
      line["verify"] = {"max_err_values": err_v, "max_err_grad": err_g,
                        "ok": bool(err_v < 1e-4 * max(len(op) for op in ops) and err_g < 1e-4)}
    if not args.no_cpu_baseline:
      # The oracle must be there when the check is asked for: a missing checker is an error of its own, never a
      # silently absent parity_check.
      from oracle import qhbm_cpu as C  # pylint: disable=import-outside-toplevel
      if not os.path.exists(C.LIB_PATH):
        line["cpu_baseline"] = {"error": f"{C.LIB_PATH} is not built (run __graft_entry__.build())"}
        line["parity_check"] = {"ok": False, "error": "the oracle library is missing: the timed workload was NOT checked"}
        parity_failed = True
      else:
        try:
          k = max(1, min(args.cpu_sample_states, spg, C.max_threads(), os.cpu_count() or 1))
          if world > 1:   # the CPU baseline is an N = 1 figure; N > 1 runs keep the parity check on a few states
            k = min(k, 8)
          bits_k = all_bits[lo:lo + k]
          # upstream 1/K: see parity_check; the rate does not depend on the weight
          oracle_params = params_np
          if os.environ.get("QHBM_BENCH_CORRUPT_PARITY") == "1":   # test hook: the check must be able to fail
            oracle_params = params_np + np.float32(0.05)
          up_k = np.tile(thetas[None, :] / float(k), (k, 1))
          rec, o_vals, o_grad, kc = cpu_baseline(n, gates, n_params, ops, oracle_params, bits_k, up_k, args.mode)
          if o_grad is not None and grad_mask is not None:
            o_grad = np.where(grad_mask, o_grad, 0.0)   # the engine returns 0 for the frozen (data) parameters
          line["cpu_baseline"] = rec if world == 1 else None
          # the checker (gate by gate) ran on the first kc of the K states, with upstream theta / kc
          bits_c, up_c = bits_k[:kc], np.tile(thetas[None, :] / float(kc), (kc, 1))
          timed_rows = vals[:kc].float().cpu().numpy()   # global rows lo..lo+kc are rank 0's own block
          grad_rows = timed_grad_rows[:kc] if timed_grad_rows is not None and timed_grad_rows.shape[0] >= kc else None
          line["parity_check"] = parity_check(eng, E, args.mode, ops, bits_c, params, timed_rows, grad_rows,
                                              float(total_states) / float(kc), up_c, o_vals, o_grad)
          parity_failed = not line["parity_check"]["ok"]
        except Exception as exc:  # pylint: disable=broad-except
          # a check that was requested and could not run is a FAILED check (exit code 3), with the reason on the line
          line.setdefault("cpu_baseline", {"error": str(exc)})
          line["parity_check"] = {"ok": False, "error": f"{type(exc).__name__}: {exc}"}
          parity_failed = True
    # "VQT step time" as BASELINE.md section 3 defines it -- loss + both gradients through vqt(), sampler excluded: the
    # same circuit and Hamiltonian through the host mirror, after everything above (the bench's engine is closed first: the
    # mirror builds its own).  A reported companion of ms_per_step (the engine's share), never the headline.
    if (world == 1 and args.mode == "vqt" and args.hamiltonian in ("xxz", "tfim") and n <= 20 and not args.no_mirror_step):
      try:
        eng.close()
        torch.cuda.empty_cache()
        line["vqt_step_through_mirror"] = mirror_step_sample(n, layers, total_states, args.hamiltonian)
        line["vqt_step_through_mirror"]["over_engine_ms_per_step"] = line["vqt_step_through_mirror"]["ms_per_step"] / ms_per_step
      except Exception as exc:  # pylint: disable=broad-except
        line["vqt_step_through_mirror"] = {"error": f"{type(exc).__name__}: {exc}"}
    print(json.dumps(line), flush=True)
    if parity_failed:
      print("bench.py: parity_check FAILED: the timed workload disagrees with the oracle: "
            f"{line['parity_check']}", file=sys.stderr)
  flag = torch.tensor([1 if (rank == 0 and parity_failed) else 0], dtype=torch.int32,
                      device="cpu" if backend == "gloo" or world == 1 else "cuda")
  if world > 1:
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
  if int(flag.item()):
    raise SystemExit(3)


if __name__ == "__main__":
  main()
