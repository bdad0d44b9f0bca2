"""Condenses gpurun_out/profile_<tag> (scripts/profile_bench.sh) into the committed summaries under
profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench command),
<tag>_bench_line.json (the bench line printed under the profiler), <tag>_counters.json (HBM traffic
and SQ counters per launch of the pass kernels), and -- with --publish -- profiles/traffic.json and
profiles/valu.json, the stored values bench.py quotes next to its live measurement.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE
are collected in separate passes (units KiB); on gfx950 FETCH_SIZE reports exactly half the bytes
of a wide (16 B per lane) coalesced read stream, which is what the tile loads are, so it is
doubled; WRITE_SIZE is taken 1:1 (calibrated on the known bytes of these kernels: each storing
pass writes every tile exactly once).  SQ_* cycle counters count quad-cycles (same guide)."""
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
publish = "--publish" in sys.argv
SRC = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
DST = os.path.join(ROOT, "profiles")
CLOCK_GHZ, SIMDS = 2.4, 1024
# Issue cost per wave instruction and SIMD, in SHADER CYCLES (scripts/experiments/micro/valu_cycles.hip, s_memtime next to
# s_memrealtime, round 5): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / DPP / v_cndmask_b32_e64 / v_readlane 4.2, plain
# 32-bit VOP1 / VOP2 2.2, v_permlane*_swap 8.1.  (Rounds 2-4 converted wall time at a nominal 2.4 GHz and read 4.6-4.8
# for the packed ops: that was the clock, 2.07 GHz under the adjoint kernel, not the instruction.)
PACKED_ISSUE_CYCLES = 4.2


def short(name):
  return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("qhbm::", "")


st = pd.read_csv(os.path.join(SRC, "stats", "bench_kernel_stats.csv"))
st["Name"] = st.Name.map(short)
st.to_csv(os.path.join(DST, f"{tag}_kernel_stats.csv"), index=False)
with open(os.path.join(SRC, "bench_line_profiled.json")) as f:
  line = json.loads(f.read())
with open(os.path.join(DST, f"{tag}_bench_line.json"), "w") as f:
  json.dump(line, f, indent=1)
head = open(os.path.join(SRC, "git_head")).read().strip()
cfg = line["config"]
meta = {"command": open(os.path.join(SRC, "command")).read().strip() + " (under rocprofv3)", "git_head": head,
        "bench_py_sha16": cfg.get("bench_py_sha16"), "kernel_sources_sha16": cfg.get("kernel_sources_sha16"), "n_qubits": cfg["n_qubits"], "layers": cfg["layers"],
        "hamiltonian": cfg.get("hamiltonian"), "mode": cfg["mode"], "states_per_gpu": cfg["states_per_gpu"]}

per = {}
for i in range(1, 9):
  path = os.path.join(SRC, f"pmc{i}", "bench_counter_collection.csv")
  if not os.path.exists(path):
    continue
  d = pd.read_csv(path)
  d = d[d.Kernel_Name.str.contains("pass_|apply_obs|observable_blocks|reduce_")].copy()
  d["Name"] = d.Kernel_Name.map(short)
  g = d.groupby(["Name", "Counter_Name"]).agg(calls=("Dispatch_Id", "nunique"), total=("Counter_Value", "sum"))
  for (name, ctr), r in g.iterrows():
    per.setdefault(name, {"calls": int(r.calls)})[ctr] = r.total / r.calls
avg_ns = {r.Name: r.AverageNs for r in st.itertuples()}
counters = dict(meta, method="per launch; hbm_bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction, see docstring)",
                kernels={})
traffic, valu = dict(meta), dict(meta)
for name, c in per.items():
  key = name.split("<")[0]
  k = dict(c, kernel=name, avg_launch_ns=avg_ns.get(name))
  if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    k["hbm_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
    if avg_ns.get(name):
      k["hbm_GBps"] = k["hbm_bytes_per_launch"] / avg_ns[name]
      k["hbm_frac_of_8TBps"] = k["hbm_GBps"] / 8000.0
    traffic[key] = {"kernel": name, "calls": c["calls"], "hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                    "FETCH_SIZE_KiB_per_launch": c["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": c["WRITE_SIZE"]}
  if "SQ_INSTS_VALU" in c and avg_ns.get(name):
    cycles = avg_ns[name] * CLOCK_GHZ  # nominal clock; GRBM_GUI_ACTIVE gives the real one when collected
    if c.get("GRBM_GUI_ACTIVE"):
      cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter is summed over the 8 XCDs
      k["effective_clock_GHz_under_profiler"] = cycles / avg_ns[name]
    k["simd_cycles_per_valu_inst"] = cycles * SIMDS / c["SQ_INSTS_VALU"]
    k["valu_issue_frac"] = c["SQ_INSTS_VALU"] * PACKED_ISSUE_CYCLES / (cycles * SIMDS)
    if "SQ_ACTIVE_INST_VALU" in c:
      k["valu_active_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * SIMDS)
    valu[key] = {"kernel": name, "valu_insts_per_launch": c["SQ_INSTS_VALU"], "valu_issue_frac": k["valu_issue_frac"],
                 "valu_active_frac": k.get("valu_active_frac"), "simd_cycles_per_valu_inst": k["simd_cycles_per_valu_inst"],
                 "effective_clock_GHz_under_profiler": k.get("effective_clock_GHz_under_profiler"),
                 "definition": "kernel cycles = GRBM_GUI_ACTIVE / 8 (measured, not a nominal clock); valu_issue_frac = "
                               "SQ_INSTS_VALU x 4.2 cycles (the issue cost of a packed-fp32 op: an UPPER bound, plain "
                               "32-bit ops cost 2.2) / (kernel cycles x 1024 SIMDs); valu_active_frac = "
                               "SQ_ACTIVE_INST_VALU x 4 / (kernel cycles x 1024 SIMDs)"}
  counters["kernels"][name] = k
with open(os.path.join(DST, f"{tag}_counters.json"), "w") as f:
  json.dump(counters, f, indent=1)
if publish:
  with open(os.path.join(DST, "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
  with open(os.path.join(DST, "valu.json"), "w") as f:
    json.dump(valu, f, indent=1)
print(st.head(8).to_string())
print(json.dumps(counters["kernels"], indent=1)[:3000])
