"""Condenses gpurun_out/profile (scripts/profile_bench.sh) into the committed summaries under
profiles/: kernel stats of the bench command and HBM traffic per launch from the PMC runs.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE
are collected in separate passes (units KiB); on gfx950 FETCH_SIZE reports exactly half the bytes
of a wide (16 B per lane) coalesced read stream, which is what the tile loads are, so it is
doubled; WRITE_SIZE is taken 1:1 after calibrating on the known bytes of these kernels (each
storing pass writes every tile exactly once)."""
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profile")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
  return name.split("(")[0].replace("void ", "").replace("qhbm::", "")


st = pd.read_csv(os.path.join(SRC, "stats", "bench_kernel_stats.csv"))
st["Name"] = st.Name.map(short)
st.to_csv(os.path.join(DST, f"{tag}_bench_kernel_stats.csv"), index=False)

with open(os.path.join(SRC, "bench_line_profiled.json")) as f:
  line = json.loads(f.read())
states = line["config"]["states_per_gpu"]

traffic = {"command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (under rocprofv3)",
           "states_per_gpu": states, "n_qubits": line["config"]["n_qubits"],
           "method": "2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 per launch (gfx950 correction, see docstring)"}
rows = {}
for ctr in ("fetch", "write"):
  d = pd.read_csv(os.path.join(SRC, ctr, "bench_counter_collection.csv"))
  d = d[d.Kernel_Name.str.contains("pass_|apply_obs")].copy()
  d["Name"] = d.Kernel_Name.map(short)
  g = d.groupby("Name").agg(calls=("Dispatch_Id", "nunique"), total=("Counter_Value", "sum"))
  for name, r in g.iterrows():
    rows.setdefault(name, {})[ctr + "_KiB_per_launch"] = r.total / r.calls
    rows[name]["calls"] = int(r.calls)
for name, r in rows.items():
  key = name.split("<")[0]
  traffic[key] = {
      "kernel": name, "calls": r["calls"],
      "FETCH_SIZE_KiB_per_launch": r["fetch_KiB_per_launch"],
      "WRITE_SIZE_KiB_per_launch": r["write_KiB_per_launch"],
      "hbm_bytes_per_launch": 2.0 * r["fetch_KiB_per_launch"] * 1024 + r["write_KiB_per_launch"] * 1024,
  }
with open(os.path.join(DST, "traffic.json"), "w") as f:
  json.dump(traffic, f, indent=1)
with open(os.path.join(DST, f"{tag}_bench_line_profiled.json"), "w") as f:
  json.dump(line, f, indent=1)
print(st.head(6).to_string())
print(json.dumps(traffic, indent=1))
