"""bench.py end to end on the GPU: the single-process line and the N > 1 launch exactly as the
driver issues it (torch.distributed.run, one process per rank).  The box has one GPU, so the two
ranks share cuda:0 and the collectives travel over gloo (bench.py's test hooks); the sharding,
barrier/max timing and aggregation logic is the code the 8-GPU run uses."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--qubits", "12", "--layers", "2", "--states-per-gpu", "16", "--steps", "2", "--warmup", "1",
         "--hamiltonian", "tfim", "--verify"]


def _line(out):
  lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, out[-2000:]
  return json.loads(lines[0])


def test_bench_single_process_line():
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-sample-states", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-2000:]
  line = _line(out.stdout)
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
    assert key in line, key
  assert line["n_gpus"] == 1 and line["steps"] == 2 and line["scaling"] == "weak"
  assert line["value"] > 0 and 0 < line["roofline"]["frac"] <= 1.0   # a fraction of the HBM peak, not a fusion factor
  assert line["roofline"]["achieved"] == pytest.approx(line["roofline"]["frac"] * line["roofline"]["peak"])
  # both ceilings are spelled out and the bound is the larger fraction (never hard-coded)
  hbm, comp = line["roofline"]["hbm"], line["roofline"]["compute"]
  assert 0 < hbm["frac"] <= 1.0 and hbm["peak"] == 8000.0 and 0 < comp["frac"] <= 1.0 and comp["peak"] == 157.3
  assert comp["achieved_TFs"] == pytest.approx(comp["frac"] * 157.3) and comp["flops_per_launch"] > 0
  # the ATTAINABLE compute ceiling is measured in the run (qhbm_clock_probe): sustained packed-fp32 rate at the clock
  # the chip holds right after the timed region -- below the nominal 157.3, above what the kernel reaches
  probe = comp["probe"]
  assert 1.0 < probe["ghz"] <= 2.6 and 3.9 <= probe["cycles_per_pk_fma"] <= 5.0, probe
  assert comp["attainable_peak"] == pytest.approx(probe["tflops"]) and 60.0 < probe["tflops"] <= 157.3
  assert comp["attainable_frac"] == pytest.approx(comp["achieved_TFs"] / probe["tflops"])
  assert line["roofline"]["bound"] == ("fp32_valu" if comp["frac"] > hbm["frac"] else "hbm")
  assert line["roofline"]["frac"] == pytest.approx(max(hbm["frac"], comp["frac"]))
  assert "12-qubit TFIM ring" in line["config"]["workload"] and line["config"]["states_total"] == 16
  # the TIMED CPU path is the diagonal-merging one; the CHECKER of parity_check stays the
  # gate-by-gate restatement, and the record says how far the two are apart on the checker's states
  cb = line["cpu_baseline"]
  assert cb["kind"] == "port" and cb["variant"] == "port+diag" and cb["value"] > 0 and cb["checker"]["states"] == 2
  assert cb["checker"]["max_diff_values_timed_path_vs_checker"] <= 2e-6
  assert cb["checker"]["max_diff_grad_timed_path_vs_checker"] <= 1e-5
  assert line["verify"]["ok"], line["verify"]
  pc = line["parity_check"]
  assert pc["ok"] and pc["states"] == 2 and pc["max_err_values"] <= pc["tol_values"]
  assert pc["max_err_grad"] <= pc["tol_grad"]
  # the gradient that is checked is the TIMED one: rows of the last timed adjoint sweep, not a second call
  assert pc["grad_from"].startswith("rows of the last timed step")
  assert line["config"]["exchange_bytes"] is None
  # BASELINE.md's "VQT step time" -- loss + both gradients through the mirror's vqt(), sampler excluded -- rides along
  through = line["vqt_step_through_mirror"]
  assert "error" not in through and through["ms_per_step"] > 0 and 1 <= through["unique_bitstrings"] <= 16
  assert through["over_engine_ms_per_step"] == pytest.approx(through["ms_per_step"] / line["ms_per_step"])


def test_bench_parameter_shift_mode_says_where_its_checked_gradient_comes_from():
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--qubits", "12", "--layers", "1",
                        "--states-total", "4", "--steps", "1", "--warmup", "0", "--hamiltonian", "tfim", "--mode", "shift",
                        "--cpu-sample-states", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-2000:]
  pc = _line(out.stdout)["parity_check"]
  assert pc["ok"] and pc["grad_from"].startswith("engine VJP of these K states")


def test_bench_config3_timed_batch_against_the_oracle():
  """BASELINE configs[2]'s circuit with bench.py's own seeds (parameters 1234, bitstrings 4321): rows of
  the timed step and the VJP of the same states against the C oracle, inside the bench run itself
  (reference pattern simulate / compare / assert, tests/inference/qnn_test.py:183-264)."""
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--states-total", "96", "--steps", "1",
                        "--warmup", "1", "--cpu-sample-states", "8"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-2000:]
  line = _line(out.stdout)
  assert (line["config"]["n_qubits"], line["config"]["layers"], line["config"]["pauli_terms"]) == (20, 16, 57)
  pc = line["parity_check"]
  assert pc["ok"] and pc["states"] == 2, pc          # (the gate-by-gate checker runs on the first 2 of the 8 timed CPU states)
  assert line["cpu_baseline"]["checker"]["max_diff_values_timed_path_vs_checker"] <= 5e-6
  assert pc["max_err_values"] <= pc["tol_values"] == pytest.approx(5e-5 * 47.5)
  assert pc["max_err_grad"] <= pc["tol_grad"] and pc["grad_inf_norm"] > 1e-2
  assert pc["grad_from"].startswith("rows of the last timed step")


def test_bench_exits_non_zero_when_the_oracle_disagrees():
  """A corrupted workload (QHBM_BENCH_CORRUPT_PARITY=1 perturbs the oracle's parameters) must fail the run."""
  env = dict(os.environ, QHBM_BENCH_CORRUPT_PARITY="1")
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-sample-states", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
  assert out.returncode != 0 and "parity_check FAILED" in out.stderr
  pc = _line(out.stdout)["parity_check"]
  assert not pc["ok"]
  # BOTH legs see the corruption: the timed values and the rows of the timed gradient
  assert pc["max_err_values"] > pc["tol_values"] and pc["max_err_grad"] > pc["tol_grad"]
  assert pc["grad_from"].startswith("rows of the last timed step")


def test_bench_fails_when_the_requested_check_cannot_run(tmp_path):
  """A parity check that was asked for and could not run (here: the oracle library is hidden) is a failed run with
  the reason on the line -- never a line that merely lacks `parity_check`."""
  env = dict(os.environ, QHBM_ORACLE_LIB=str(tmp_path / "missing.so"))
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-sample-states", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
  assert out.returncode != 0
  line = _line(out.stdout)
  assert line["parity_check"]["ok"] is False and "error" in line["parity_check"]


@pytest.mark.parametrize("reduction", ["allreduce", "ordered"])
def test_bench_two_ranks_as_the_driver_launches_it(reduction):
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  env = dict(os.environ, QHBM_BENCH_SHARE_DEVICE="1", QHBM_BENCH_BACKEND="gloo")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
         "--gpus", "2", "--cpu-sample-states", "2", "--reduction", reduction] + SMALL
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
  assert out.returncode == 0, out.stderr[-3000:]
  line = _line(out.stdout)
  assert line["n_gpus"] == 2
  assert line["config"]["parallelism"] == "batch-sharded x2"
  # what the collective backend reports, and the exchange that ran (ordered = the host mirror's default)
  assert line["config"]["backend"] == "gloo" and line["config"]["backend_world_size"] == 2
  assert line["config"]["reduction"] == reduction and len(line["config"]["devices"]) >= 1
  assert line["verify"]["ok"], line["verify"]
  assert line["parity_check"]["ok"], line["parity_check"]
  # bytes of the step's exchange: values [32] + the [P] gradient, or + the rows [32, P]
  n_params = 2 * (3 * 12 - 1)
  assert line["config"]["exchange_bytes"] == 4 * 32 + 4 * (32 * n_params if reduction == "ordered" else n_params)
  # every rank's own kernel time and block: what diagnoses a straggler in the driver's scaling run
  pr = line["per_rank"]
  assert len(pr["kernel_ms_per_step"]) == 2 and pr["states"] == [16, 16] and pr["balance"] == "equal"
  assert 0 < pr["min"] <= pr["max"] == pr["kernel_ms_per_step"][pr["argmax"]]


def test_bench_eight_ranks_as_the_driver_launches_it():
  """The 8-rank code path of the round-end scaling run (no 8-GPU node here: eight gloo ranks share the one device, tiny
  size): 36 states split 5 + 5 + 5 + 5 + 4 + 4 + 4 + 4, --verify against rank 0's own evaluation of the whole batch,
  the exchange bytes and what the backend reports about its world."""
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  env = dict(os.environ, QHBM_BENCH_SHARE_DEVICE="1", QHBM_BENCH_BACKEND="gloo")
  args = ["--qubits", "12", "--layers", "2", "--states-total", "36", "--steps", "2", "--warmup", "1",
          "--hamiltonian", "tfim", "--cpu-sample-states", "2"]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8"] + args
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
  assert out.returncode == 0, out.stderr[-3000:]
  line = _line(out.stdout)
  assert line["n_gpus"] == 8 and line["scaling"] == "strong"
  assert line["config"]["parallelism"] == "batch-sharded x8"
  assert line["config"]["backend"] == "gloo" and line["config"]["backend_world_size"] == 8
  assert line["config"]["states_total"] == 36 and line["config"]["states_per_gpu"] == 5
  assert line["config"]["reduction"] == "allreduce"           # the default: [P] floats, not the [U, P] rows
  n_params = 2 * (3 * 12 - 1)
  assert line["config"]["exchange_bytes"] == 4 * 36 + 4 * n_params
  assert line["verify"]["ok"], line["verify"]                 # on by default for N > 1
  assert line["parity_check"]["ok"], line["parity_check"]
  pr = line["per_rank"]
  assert pr["states"] == [5, 5, 5, 5, 4, 4, 4, 4] and len(pr["kernel_ms_per_step"]) == 8 and pr["weights"] is None


def test_bench_balance_measured_deals_blocks_by_measured_speed():
  """`--balance measured` (three gloo ranks sharing the one device): a probing step, every rank's kernel time per state
  all-gathered once (parallel.measured_weights), blocks proportional to the speeds; the sharded result still equals rank
  0's own evaluation of the whole batch and the oracle."""
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  env = dict(os.environ, QHBM_BENCH_SHARE_DEVICE="1", QHBM_BENCH_BACKEND="gloo")
  args = ["--qubits", "12", "--layers", "2", "--states-total", "90", "--steps", "2", "--warmup", "1",
          "--hamiltonian", "tfim", "--cpu-sample-states", "2", "--balance", "measured"]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "3"] + args
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
  assert out.returncode == 0, out.stderr[-3000:]
  line = _line(out.stdout)
  pr = line["per_rank"]
  assert pr["balance"] == "measured" and len(pr["weights"]) == 3 and all(w > 0 for w in pr["weights"])
  assert sum(pr["states"]) == 90 and min(pr["states"]) >= 1
  total = sum(pr["weights"])
  assert all(abs(s - 90 * w / total) <= 1.0 for s, w in zip(pr["states"], pr["weights"]))
  assert line["verify"]["ok"] and line["parity_check"]["ok"]


def test_bench_gpus_flag_starts_its_own_ranks_and_shards_a_fixed_total():
  """`bench.py --gpus 2` outside torchrun launches two ranks itself (strong scaling: 33 states split
  17 + 16); on this one-GPU box that needs the share-device test hook, and WITHOUT it the run must
  refuse rather than print an `n_gpus: 1` line."""
  strong = ["--qubits", "12", "--layers", "2", "--states-total", "33", "--steps", "2", "--warmup", "1",
            "--hamiltonian", "tfim", "--verify", "--no-cpu-baseline", "--gpus", "2"]
  env = dict(os.environ, QHBM_BENCH_SHARE_DEVICE="1", QHBM_BENCH_BACKEND="gloo")
  env.pop("WORLD_SIZE", None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + strong, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
  assert out.returncode == 0, out.stderr[-3000:]
  line = _line(out.stdout)
  assert line["n_gpus"] == 2 and line["scaling"] == "strong"
  assert line["config"]["states_total"] == 33 and line["config"]["states_per_gpu"] == 17
  assert line["verify"]["ok"], line["verify"]
  import torch
  if torch.cuda.device_count() < 2:
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "QHBM_BENCH_SHARE_DEVICE", "QHBM_BENCH_BACKEND"):
      env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + strong, capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "GPU(s) visible" in out.stderr
    assert '{"metric"' not in out.stdout


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip("one GPU visible: the nccl (RCCL) variant needs two")
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-3000:]
  line = _line(out.stdout)
  assert line["n_gpus"] == 2 and line["verify"]["ok"], line


def test_bench_qmhl_mode_checks_the_masked_gradient_of_the_timed_step():
  """`--mode qmhl`: U_data then U_model^dagger, the KOBE-2 shards of the model as observables, the adjoint VJP with
  respect to the model's parameters only (qmhl_loss.py:33-34, qnn.py:120-139).  Values [U, T] and the gradient rows of
  the timed step against the C oracle; the frozen (data) half of the gradient is exactly zero."""
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "qmhl", "--qubits", "13", "--layers", "2",
                        "--states-total", "24", "--steps", "1", "--warmup", "1", "--cpu-sample-states", "4"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-2000:]
  line = _line(out.stdout)
  cfg = line["config"]
  assert cfg["mode"] == "qmhl" and cfg["observables"] == 13 + 13 * 12 // 2 and cfg["hamiltonian"] == "kobe2_shards"
  assert line["qmhl_step_ms"] == line["ms_per_step"] and line["vqt_step_ms"] is None
  pc = line["parity_check"]
  assert pc["ok"] and pc["states"] == 2 and pc["grad_from"].startswith("rows of the last timed step"), pc
  assert pc["max_err_values"] <= pc["tol_values"] == pytest.approx(5e-5) and pc["grad_inf_norm"] > 1e-3


def test_bench_three_observables_in_one_call():
  """`--hamiltonian xxz3`: the XXZ chain as its XX, YY and ZZ sums (several operators per call, the reference's
  normal usage, tests/inference/qnn_test.py:187-190): values of the timed step per observable and the VJP against
  the oracle; the three values come from the observable kernel (one launch), not from measuring passes."""
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--hamiltonian", "xxz3", "--qubits", "14",
                        "--layers", "3", "--states-total", "16", "--steps", "1", "--warmup", "1", "--cpu-sample-states", "4"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-2000:]
  line = _line(out.stdout)
  assert line["config"]["observables"] == 3 and line["config"]["pauli_terms"] == 3 * 13
  assert line["parity_check"]["ok"], line["parity_check"]
  assert line["kernel_ms_per_step"]["apply_observable"] > 0


def test_bench_names_the_block_kernel_when_the_pauli_sum_has_hundreds_of_masks():
  """Config 4's operator (512 random strings) on 16 qubits: lambda = O psi is the largest kernel of the step and
  `roofline.kernel` names the kernel that ran -- the block-grouped one (engine.cpp block_kernel), not the family label."""
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--qubits", "16", "--layers", "2", "--hamiltonian", "random512",
                        "--states-total", "8", "--steps", "1", "--warmup", "1", "--cpu-sample-states", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-2000:]
  line = _line(out.stdout)
  assert line["parity_check"]["ok"], line["parity_check"]
  assert line["config"]["pauli_terms"] == 512
  k = line["kernel_ms_per_step"]
  if k["apply_observable"] > max(k["forward"], k["adjoint"]):
    assert line["roofline"]["kernel"] == "observable_blocks_kernel", line["roofline"]
