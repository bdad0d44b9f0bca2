"""Loading of tests/golden/*.npz (see tests/golden/make_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
  return dict(np.load(os.path.join(GOLDEN, name)))


def gates_of(arr):
  return [(int(k), int(q0), int(q1), int(p), float(s), float(o)) for k, q0, q1, p, s, o in arr]


def ops_of(arr):
  n_ops = int(arr[:, 0].max()) + 1 if len(arr) else 0
  ops = [[] for _ in range(n_ops)]
  for k, c, x, z in arr:
    ops[int(k)].append((float(c), int(x), int(z)))
  return ops


def hea_files():
  return sorted(f for f in os.listdir(GOLDEN) if f.startswith("hea_n") and "bit_order" not in f)
