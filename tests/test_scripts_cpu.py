"""Developer scripts that must not rot: the ablation builder's text anchors into kernels.hip
(scripts/experiments/ablate/build.py builds the variants HISTORY.md's ablation tables quote)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_ablation_variant_still_applies_to_the_kernel_source():
  spec = importlib.util.spec_from_file_location("ablate_build", os.path.join(ROOT, "scripts", "experiments", "ablate", "build.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  assert len(mod.VARIANTS) >= 25
  assert mod.check_variants() == {}
