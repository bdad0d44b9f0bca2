"""Shared test setup: paths, the `gpu` marker, and the engine fixture."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "qhbm-library_amd")
for p in (ROOT, SRC):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line(
      "markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_available():
  try:
    import torch
    return torch.cuda.is_available()
  except Exception:  # pylint: disable=broad-except
    return False


def pytest_collection_modifyitems(config, items):
  # `-m gpu` on a box without a GPU must not silently pass.
  del config
  if _gpu_available():
    return
  skip = pytest.mark.skip(reason="no GPU visible")
  for item in items:
    if "gpu" in item.keywords:
      item.add_marker(skip)
