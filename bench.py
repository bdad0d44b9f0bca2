#!/usr/bin/env python3
"""Benchmark of the expectation hot path (BASELINE.json metric).

One "step" = one VQT step's engine work over one batch of synthetic input:
values [U, 1] of the target Hamiltonian AND the adjoint vector-Jacobian
product (the [P] gradient of the VQT loss w.r.t. the circuit parameters), for
U EBM bitstrings already resident in HBM.  With --gpus N the batch is sharded
over N ranks (one process per GPU, weak scaling: --states-per-gpu is fixed)
and each step ends with the exchange the path really has: an RCCL all-reduce
of the [P] gradient and an all-gather of the per-state expectations.

  value = (states over all ranks) x (Pauli terms) / step time      [evals/s]

Workload (config.workload): BASELINE.json configs[2] -- 20-qubit XXZ chain,
depth-16 hardware-efficient ansatz, 4096 samples sharded over 8 GPUs, i.e.
512 states per GPU; it fits one GPU, so the same per-GPU shard is the N=1
workload.  Synthetic inputs: phi ~ U[-1,1] seeded, distinct
seeded bitstrings (U = B exactly).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "qhbm-library_amd")):
  if _p not in sys.path:
    sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


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


def cpu_baseline(n, gates, n_params, op, params, sample_states, mode):
  """The oracle's C restatement (oracle/qhbm_cpu.c) timed on this host's cores on
  a bounded sample of the same workload -- a reported baseline, never the product."""
  from oracle import qhbm_cpu as C
  if not os.path.exists(C.LIB_PATH):
    return None
  cores = min(C.max_threads(), os.cpu_count() or 1)
  states = max(1, min(sample_states, cores))
  bits = distinct_bitstrings(n, states, 999)
  up = np.full((states, 1), 1.0 / states, np.float32)
  t0 = time.perf_counter()
  if mode == "forward":
    C.expectation(n, gates, params, bits, [op], n_threads=cores)
  else:
    C.expectation_vjp(n, gates, params, bits, [op], up, n_threads=cores)
  dt = time.perf_counter() - t0
  return {
      "value": states * len(op) / dt, "unit": "evals/s", "cores": int(min(cores, states)),
      "kind": "port",
      "sample": f"{states} states of the same workload ({mode} step), one state per thread, "
                f"{dt:.2f} s wall",
  }


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--qubits", type=int, default=20)
  ap.add_argument("--layers", type=int, default=16)
  ap.add_argument("--states-per-gpu", type=int, default=512)
  ap.add_argument("--hamiltonian", choices=["xxz", "tfim", "random512"], default="xxz")
  ap.add_argument("--mode", choices=["vqt", "forward", "shift"], default="vqt")
  ap.add_argument("--tile-qubits", type=int, default=0)
  ap.add_argument("--adjoint-tile-qubits", type=int, default=0)
  ap.add_argument("--cpu-sample-states", type=int, default=64)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--verify", action="store_true",
                  help="rank 0 re-evaluates the whole batch alone and compares with the sharded result")
  args = ap.parse_args()

  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus and world > 1:
    args.gpus = world
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
  # Test hooks (tests/test_bench_gpu.py runs two ranks on ONE GPU): QHBM_BENCH_SHARE_DEVICE=1 puts
  # every rank on cuda:0 and QHBM_BENCH_BACKEND=gloo carries the collectives through the host.
  if os.environ.get("QHBM_BENCH_SHARE_DEVICE") == "1":
    local_rank = 0
  backend = os.environ.get("QHBM_BENCH_BACKEND", "nccl")
  torch.cuda.set_device(local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)

  def all_reduce_sum(t):
    if backend == "gloo":
      c = t.cpu()
      dist.all_reduce(c)
      t.copy_(c)
    else:
      dist.all_reduce(t)

  def all_gather(outs, t):
    if backend == "gloo":
      couts = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
      dist.all_gather(couts, t.cpu())
      for o, c in zip(outs, couts):
        o.copy_(c)
    else:
      dist.all_gather(outs, t)

  from qhbmlib_amd import _engine as E  # pylint: disable=import-outside-toplevel

  n, layers, spg = args.qubits, args.layers, args.states_per_gpu
  gates, n_params = hea_gates(n, layers)
  op = {"xxz": xxz_op, "tfim": tfim_op, "random512": lambda m: random_pauli_op(m, 512, 24)}[args.hamiltonian](n)
  rng = np.random.default_rng(1234)
  params_np = rng.uniform(-1, 1, n_params).astype(np.float32)
  all_bits = distinct_bitstrings(n, spg * world, 4321)
  bits = torch.from_numpy(all_bits[rank * spg:(rank + 1) * spg]).cuda()
  params = torch.from_numpy(params_np).cuda()
  total_states = spg * world
  upstream = torch.full((spg, 1), 1.0 / total_states, device="cuda")

  eng = E.Engine(local_rank)
  if args.tile_qubits:
    eng.set_option("tile_qubits", args.tile_qubits)
  if args.adjoint_tile_qubits:
    eng.set_option("adjoint_tile_qubits", args.adjoint_tile_qubits)
  eng.set_circuit(n, gates, n_params)
  eng.set_observables([op])
  eng.set_option("profile_events", 1)
  fwd_passes, bwd_passes = eng.num_passes()
  gathered = [torch.empty((spg, 1), device="cuda") for _ in range(world)] if world > 1 else None

  def step():
    if args.mode == "forward":
      vals = eng.expectation(bits, params)
      grad = None
    else:
      vals, grad = eng.expectation_vjp(bits, params, upstream,
                                       method=E.GRAD_PARAMETER_SHIFT if args.mode == "shift" else E.GRAD_ADJOINT)
    if world > 1:
      if grad is not None:
        all_reduce_sum(grad)
      all_gather(gathered, vals)
    return vals, grad

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

  if rank == 0:
    n_terms = len(op)
    evals_per_step = total_states * n_terms
    ms_per_step = dt / args.steps * 1e3
    # ---- roofline of the dominant kernel (algorithmic bytes) ----
    n_gate = len(gates)
    amp = float(1 << n)
    fwd_alg = spg * (16.0 * n_gate + 8.0 * n_terms + 8.0) * amp  # per step, this rank
    if args.mode == "shift":  # one base forward + 2 P shifted forwards
      fwd_alg *= 1 + 2 * n_params
    bwd_alg = spg * 48.0 * n_gate * amp
    n_diag = sum(1 for g in gates if g[0] in (E.GATE_ZPOW, E.GATE_CZPOW, E.GATE_ZZPOW))
    fwd_flops = spg * amp * (14.0 * (n_gate - n_diag) + 6.0 * n_diag)
    use_bwd = args.mode == "vqt" and kt["bwd_ms"] >= kt["fwd_ms"]
    if use_bwd:
      launches, ms, alg, name = kt["bwd_launches"], kt["bwd_ms"], bwd_alg, "pass_adj_kernel"
    else:
      launches, ms, alg, name = kt["fwd_launches"], kt["fwd_ms"], fwd_alg, "pass_fwd_kernel"
    alg_flops = 3.0 * fwd_flops if use_bwd else fwd_flops * ((1 + 2 * n_params) if args.mode == "shift" else 1)
    per_step_launches = max(1, launches // max(1, args.steps))
    avg_ms = ms / max(1, launches)
    achieved = (alg / per_step_launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # physical HBM traffic of one launch if every tile is read and written once
    io_bytes = spg * amp * 8.0 * 2.0 * (2.0 if use_bwd else 1.0)
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3
    # passes of this same command; scripts/profile_bench.sh + scripts/summarize_profile.py).
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
      try:
        with open(tpath) as f:
          tj = json.load(f)
        if tj.get("states_per_gpu") == spg and tj.get("n_qubits") == n and args.hamiltonian == "xxz":
          traffic = tj.get(name, {}).get("hbm_bytes_per_launch")
      except Exception:  # pylint: disable=broad-except
        traffic = None
    line = {
        "metric": "circuit-expectation evals/sec (samples×Pauli terms) at n qubits; VQT step time",
        "value": evals_per_step / (dt / args.steps),
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[2] shard: {n}-qubit "
                         f"{ {'xxz': 'XXZ(delta=0.5) open chain', 'tfim': 'TFIM ring', 'random512': 'random 512-term Pauli sum'}[args.hamiltonian]}, "
                         f"HEA depth {layers} ({n_params} params), {spg} states/GPU, "
                         f"{ {'vqt': 'VQT step = values + adjoint VJP', 'forward': 'forward values only', 'shift': 'values + parameter-shift VJP'}[args.mode]}"),
            "n_qubits": n, "layers": layers, "states_per_gpu": spg, "pauli_terms": n_terms,
            "mode": args.mode, "parallelism": f"batch-sharded x{world}",
            "forward_passes": fwd_passes, "adjoint_passes": bwd_passes,
        },
        "vqt_step_ms": ms_per_step if args.mode == "vqt" else None,
        "kernel_ms_per_step": {"forward": kt["fwd_ms"] / args.steps, "adjoint": kt["bwd_ms"] / args.steps},
        "roofline": {
            "bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "avg_launch_ms": avg_ms, "launches_per_step": per_step_launches,
            "algorithmic_bytes_per_launch": alg / per_step_launches,
            "tile_io_bytes_per_launch": io_bytes,
            "tile_io_GBps": io_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
            # the fused kernels are VALU-bound, so the same launch priced in SURVEY.md 8(d)'s
            # algorithmic flops (28 per amplitude pair for a one-qubit gate, 6 per amplitude for a
            # diagonal one; the adjoint sweeps every gate three times) against the fp32 vector peak
            "algorithmic_TFLOPs": alg_flops / per_step_launches / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
            "fp32_vector_peak_TFLOPs": 157.3,
        },
    }
    if args.verify:
      # the sharded step against one process evaluating every state (same engine, same inputs)
      full_bits = torch.from_numpy(all_bits).cuda()
      full_up = torch.full((total_states, 1), 1.0 / total_states, device="cuda")
      if args.mode == "forward":
        ref_vals, ref_grad = eng.expectation(full_bits, params), None
      else:
        ref_vals, ref_grad = eng.expectation_vjp(
            full_bits, params, full_up,
            method=E.GRAD_PARAMETER_SHIFT if args.mode == "shift" else E.GRAD_ADJOINT)
      got_vals = torch.cat(gathered) if world > 1 else vals
      err_v = float((got_vals - ref_vals).abs().max())
      err_g = float((grad - ref_grad).abs().max()) if ref_grad is not None else 0.0
      line["verify"] = {"max_err_values": err_v, "max_err_grad": err_g,
                        "ok": bool(err_v < 1e-4 * len(op) and err_g < 1e-4)}
    if not args.no_cpu_baseline:
      try:
        line["cpu_baseline"] = cpu_baseline(n, gates, n_params, op, params_np, args.cpu_sample_states, args.mode)
      except Exception as exc:  # pylint: disable=broad-except
        line["cpu_baseline"] = {"error": str(exc)}
    print(json.dumps(line))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
