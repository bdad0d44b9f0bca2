"""Register / LDS / occupancy table of the pass and observable kernels as hipcc reports them, and WHERE their spills sit
(developer tool and CPU test, no GPU needed):
    python scripts/kernel_resources.py            # the table, then every spill site with its loop context
`check()` is what tests/test_scripts_cpu.py runs: a kernel the DEFAULT planner can select must not spill a VGPR at all and
must not spill / reload an SGPR inside its instance loop (round 5's review, item 4 b).

How a spill is found in the ISA (hipcc -S, gfx950): VGPR spills are the scratch accesses LLVM annotates "Folded Spill" /
"Folded Reload"; SGPR spills are v_writelane_b32 / v_readlane_b32 pairs on a VGPR the kernel uses for nothing else (the pass
kernels read record fields with scalar loads, never with lane reads).  The INSTANCE LOOP of a pass kernel is the smallest
loop (label ... backward branch) that holds at least 90 % of the kernel's packed-fp32 instructions; a spill inside it costs
a 4.2-cycle VALU slot per instance and wave on a port that is 0.93 busy, one outside it a few cycles per round or per
workgroup."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qhbm-library_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-promote-alloca-to-vector=1",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp", "--cuda-device-only"]   # = csrc/Makefile CXXFLAGS + KFLAGS
SOURCES = ("kernels.hip", "observable.hip")


def _demangle(name):
  out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
  return out.replace("(anonymous namespace)::", "").split("(")[0].replace("void qhbm::", "").strip()


def resource_rows():
  """One dict per kernel: name, VGPRs, SGPRs, VGPR / SGPR spill counts, occupancy (-Rpass-analysis=kernel-resource-usage)."""
  out = ""
  for src in SOURCES:
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out += subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
  rows, cur = [], None
  for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
      cur = {"mangled": m.group(1)}
      rows.append(cur)
      continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
      cur[m.group(1).strip()] = int(m.group(2))
  for r in rows:
    r["name"] = _demangle(r["mangled"])
  return [r for r in rows if any(k in r["name"] for k in ("pass_", "apply_obs", "observable_blocks"))]


def assembly(src):
  cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", src, "-o", "-"]
  return subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stdout


def functions(asm):
  """{mangled name: list of body lines}."""
  out, cur, body = {}, None, []
  for line in asm.splitlines():
    m = re.match(r"^(_Z\S+):\s+; @", line)
    if m:
      cur, body = m.group(1), []
      continue
    if cur is not None:
      if line.startswith(".Lfunc_end"):
        out[cur] = body
        cur = None
      else:
        body.append(line)
  return out


def loops(body):
  """[(first line, last line)] of every label ... backward-branch range."""
  labels = {}
  for i, line in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
      labels[m.group(1)] = i
  found = []
  for i, line in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
      found.append((labels[m.group(1)], i))
  return found


_PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


def instance_loop(body):
  """The smallest loop with >= 90 % of the kernel's packed-fp32 instructions (None: a kernel without such a loop)."""
  packed = [i for i, line in enumerate(body) if _PACKED.search(line)]
  if len(packed) < 32:
    return None
  best = None
  for lo, hi in loops(body):
    inside = sum(1 for i in packed if lo <= i <= hi)
    if inside >= 0.9 * len(packed) and (best is None or hi - lo < best[1] - best[0]):
      best = (lo, hi)
  return best


def spill_sites(body):
  """[(kind, line index, text)]: kind in {"vgpr spill", "vgpr reload", "sgpr spill", "sgpr reload"}."""
  sites = []
  lane_regs = set()
  for line in body:
    m = re.search(r"\bv_writelane_b32\s+(v\d+),\s*s\d+,\s*\d+", line)
    if m:
      lane_regs.add(m.group(1))
  for i, line in enumerate(body):
    text = line.strip()
    if "Folded Spill" in line:
      sites.append(("vgpr spill", i, text))
    elif "Folded Reload" in line:
      sites.append(("vgpr reload", i, text))
    else:
      m = re.search(r"\bv_writelane_b32\s+(v\d+),\s*s\d+,\s*\d+", line)
      if m and m.group(1) in lane_regs:
        sites.append(("sgpr spill", i, text))
      m = re.search(r"\bv_readlane_b32\s+s\d+,\s*(v\d+),\s*\d+", line)
      if m and m.group(1) in lane_regs:
        sites.append(("sgpr reload", i, text))
  return sites


def default_selectable(name):
  """Kernels the planner picks WITHOUT options: lean pass kernels (every gate kind is lowered to X powers and phases, so
  the GENERAL variants -- `<..., true>` -- run only under `force_general_kernels`), tiles of 2^10 .. 2^14, both row modes of
  the exchange adjoint kernel, and the observable kernels but the far launches of the two-level sweep and the block kernel's
  shape for blocks of 2^12."""
  if name.startswith("pass_fwd_kernel") or name.startswith("pass_adj_kernel"):
    return name.endswith("false>")
  if name.startswith("apply_observable_kernel"):
    return name.endswith("false>")   # (`<..., true>`: the far launches of the two-level sweep, `observable_far_windows`, default off)
  if name.startswith("observable_blocks_kernel"):
    return name.endswith(", 13>")    # (`<..., 12>`: the two-workgroups-per-CU shape, option "observable_block_bits" = 12)
  return name.startswith(("pass_fwd2_kernel", "pass_adjx_kernel"))


def report():
  """(rows, sites): sites[name] = [(kind, line, in_instance_loop, text)]."""
  rows = resource_rows()
  by_mangled = {}
  for src in SOURCES:
    by_mangled.update(functions(assembly(src)))
  sites = {}
  for r in rows:
    body = by_mangled.get(r["mangled"])
    if body is None:
      continue
    hot = instance_loop(body)
    r["instance_loop"] = hot
    sites[r["name"]] = [(kind, i, hot is not None and hot[0] <= i <= hot[1], text) for kind, i, text in spill_sites(body)]
  return rows, sites


def check():
  """Violations (strings) of the two rules for default-selectable kernels; empty = fine."""
  rows, sites = report()
  bad = check_from(rows, sites)
  seen = sum(1 for r in rows if default_selectable(r["name"]))
  if seen < 20:
    bad.append(f"only {seen} default-selectable kernels found: the name filter no longer matches the sources")
  return bad


def main():
  rows, sites = report()
  for r in rows:
    print(f"{r['name']:38s} VGPR {r.get('VGPRs', -1):4d}  SGPR {r.get('TotalSGPRs', -1):4d}  spill v{r.get('VGPRs Spill', 0)}/s"
          f"{r.get('SGPRs Spill', 0)}  waves/SIMD {r.get('Occupancy', -1)}  {'default' if default_selectable(r['name']) else 'option-only'}")
  print()
  for r in rows:
    ss = sites.get(r["name"], [])
    if not ss:
      continue
    hot = r.get("instance_loop")
    inside = sum(1 for s in ss if s[2])
    print(f"{r['name']}: instance loop = lines {hot}, {len(ss)} spill / reload instructions, {inside} inside the instance loop")
    for kind, i, in_hot, text in ss:
      print(f"    line {i:5d}  {kind:11s} {'IN THE INSTANCE LOOP' if in_hot else ''}  {text[:100]}")
  bad = check_from(rows, sites)
  print("\nviolations:", bad if bad else "none")
  return 1 if bad else 0


def check_from(rows, sites):
  bad = []
  for r in rows:
    if not default_selectable(r["name"]):
      continue
    if r.get("VGPRs Spill", 0):
      bad.append(f"{r['name']}: {r['VGPRs Spill']} VGPRs spilled")
    if any(s[2] for s in sites.get(r["name"], [])):
      bad.append(f"{r['name']}: spill inside the instance loop")
  return bad


if __name__ == "__main__":
  sys.exit(main())
