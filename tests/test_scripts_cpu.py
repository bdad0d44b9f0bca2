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


def test_the_block_observable_kernels_compile_without_register_spills():
  """csrc/observable.hip keeps two prefetch sets per thread as PENDING loads the compiler does not know about (volatile
  asm, manual s_waitcnt): a spilled register next to them would be stored before its data has landed.  Every mode of
  the kernel must fit its 128 registers (four waves per SIMD at 1024 threads) with no spill and no scratch -- an
  experiment that pushed the several-observables mode to 12 spills failed parity on the GPU (HISTORY.md round 4)."""
  import re
  import shutil
  import subprocess
  import pytest
  hipcc = "/opt/rocm/bin/hipcc"
  if not shutil.which(hipcc):
    pytest.skip("no hipcc")
  csrc = os.path.join(ROOT, "qhbm-library_amd", "csrc")
  out = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-promote-alloca-to-vector=1",
                        "-mllvm", "-amdgpu-sched-strategy=max-ilp", "--cuda-device-only", "-c", "observable.hip", "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True, timeout=600).stderr
  blocks = re.split(r"remark: Function Name: ", out)[1:]
  seen = 0
  for b in blocks:
    if "observable_blocks_kernel" not in b.split()[0]:
      continue
    seen += 1
    field = lambda name: int(re.search(name + r"[^:]*: (\d+)", b).group(1))
    assert field("VGPRs Spill") == 0 and field("SGPRs Spill") == 0 and field("ScratchSize") == 0, b[:400]
    assert field("VGPRs") <= 128 and field("Occupancy") >= 4, b[:400]
  assert seen == 12   # four modes x the three shapes (blocks of 2^13 split by masks / by rows, blocks of 2^12)


def test_published_counter_profiles_carry_the_hash_of_the_kernels_they_were_taken_on():
  """profiles/traffic.json and valu.json are stored values that bench.py prints next to its own measurements: they carry
  the hash of the engine sources they were taken on, and bench.py flags a mismatch in its line (`stored_profile_warning`).
  A mismatch here is reported as a warning, not a failure: the kernels may have moved on after the last profile."""
  import json
  import sys
  import warnings
  sys.path.insert(0, ROOT)
  import bench
  sha = bench.kernel_sources_sha16()
  assert sha and len(sha) == 16 and sha == bench.kernel_sources_sha16()
  for name in ("traffic.json", "valu.json"):
    with open(os.path.join(ROOT, "profiles", name)) as f:
      stored = json.load(f)
    assert len(stored.get("kernel_sources_sha16") or "") == 16, f"profiles/{name} does not say which kernels it was taken on"
    if stored["kernel_sources_sha16"] != sha:
      warnings.warn(f"profiles/{name} was taken on other engine sources ({stored['kernel_sources_sha16']}) than this tree's "
                    f"({sha}): bench.py will say so in its line; re-run scripts/r05_profiles.sh + summarize_profile.py --publish")


def test_no_register_spills_in_the_kernels_the_default_planner_selects():
  """scripts/kernel_resources.py on the shipped sources (hipcc cross-compiles: no GPU): a kernel the planner can pick
  without options spills NO VGPR (round 5 shipped 4 - 13 in `pass_adjx_kernel<10, *>`, `<11, 0>`, `<12, 2>`, `<13, 2>`:
  the store epilogue's two tiles were interleaved by the scheduler) and spills / reloads no SGPR inside its instance loop
  (the 8 - 12 SGPR lane spills of the exchange adjoint kernel sit in the workgroup's prologue and store epilogue)."""
  spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  assert mod.default_selectable("pass_adjx_kernel<12, 0>") and not mod.default_selectable("pass_adj_kernel<12, true>")
  assert mod.check() == []
