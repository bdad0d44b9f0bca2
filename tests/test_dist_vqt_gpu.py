"""A whole sharded training step on the GPU (reference loop ebm.py:271-280: ONE sample
set, dedup, then the hot path): `vqt()` + `backward()` with a sampler built with `initial_seed=None`
on 1, 2 and 3 ranks whose global torch generators are seeded differently.  The agreed sampler seed
makes every rank draw the same samples; loss and both gradients must then be BIT-IDENTICAL for any
number of ranks (single observable: the values-from-lambda path with the ordered row reduction).
Ranks that do sample different sets must fail loudly, not all-gather garbage."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, out_path, **env):
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_vqt_worker.py"),
         out_path]
  return subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                        env=dict(os.environ, QHBM_TEST_BACKEND="gloo", **env))


def test_vqt_training_steps_are_bit_identical_on_1_2_3_ranks(tmp_path):
  runs = []
  for w in (1, 2, 3):
    out = _launch(w, str(tmp_path / f"w{w}.npz"))
    assert out.returncode == 0, out.stderr[-3000:]
    runs.append(dict(np.load(str(tmp_path / f"w{w}.npz"))))
  assert [int(r["world"]) for r in runs] == [1, 2, 3]
  for r in runs[1:]:
    assert int(r["first_seed"]) == int(runs[0]["first_seed"])      # rank 0's draw (global seed 100) everywhere
    for key in ("loss0", "g_phi0", "g_theta0", "loss1", "g_phi1", "g_theta1"):
      np.testing.assert_array_equal(r[key], runs[0][key], err_msg=key)          # atol = 0
  assert np.abs(runs[0]["g_phi0"]).max() > 1e-3 and np.abs(runs[0]["g_theta0"]).max() > 1e-4
  assert not np.array_equal(runs[0]["g_phi0"], runs[0]["g_phi1"])   # the second step drew new samples


def test_ranks_that_sample_different_bitstrings_fail_loudly(tmp_path):
  out = _launch(2, str(tmp_path / "bad.npz"), QHBM_TEST_DESYNC="1")
  assert out.returncode != 0
  assert out.stderr.count("ShardMismatchError") >= 2, out.stderr[-3000:]       # raised on every rank
  assert not os.path.exists(str(tmp_path / "bad.npz"))
