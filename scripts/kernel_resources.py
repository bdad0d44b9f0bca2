"""Register / LDS / occupancy table of the pass and observable kernels as hipcc reports them (developer tool, CPU only):
    python scripts/kernel_resources.py"""
import os
import re
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qhbm-library_amd", "csrc")
out = ""
for src in ("kernels.hip", "observable.hip"):
  cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-promote-alloca-to-vector=1",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp", "--cuda-device-only", "-c", src, "-o", "/dev/null",
         "-Rpass-analysis=kernel-resource-usage"]
  out += subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
  m = re.search(r"Function Name: (\S+)", line)
  if m:
    cur = {"name": m.group(1)}
    rows.append(cur)
    continue
  m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
  if m and cur is not None:
    cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
  name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.replace("(anonymous namespace)::", "").split("(")[0].replace("void qhbm::", "")
  if "pass_" not in name and "apply_obs" not in name and "observable_blocks" not in name:
    continue
  print(f"{name:36s} VGPR {r.get('VGPRs', -1):4d}  SGPR {r.get('TotalSGPRs', -1):4d}  spill v{r.get('VGPRs Spill', 0)}/s"
        f"{r.get('SGPRs Spill', 0)}  waves/SIMD {r.get('Occupancy', -1)}")
