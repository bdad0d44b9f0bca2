"""Dynamic instruction mix of config 3's adjoint sweep, piece by piece (VERDICT r4 #1b).

Inputs (both produced on the GPU box by `scripts/r05_ablate.sh <states> head no_reduce no_butterfly ...`):
  gpurun_out/r05_ablate/<variant>/c/**/t_counter_collection.csv   SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE per launch
  gpurun_out/r05_ablate/<variant>/c/**/t_kernel_trace.csv        launch durations of the same run
and, computed here without a device, the engine's micro-op census of the plan (qhbm_op_census).

A variant is the shipped kernel with ONE piece compiled out (scripts/experiments/ablate/build.py; its results are wrong, its
counters exact): head - variant = the instructions and the cycles of that piece.  Dividing by the census -- how many
times a wave executes that micro-op per state -- gives instructions per execution, next to the packed-fp32 operations
the formulation needs for it (the count of the inline-asm sequences in csrc/kernels.hip).

  python scripts/instruction_mix.py [states] > profiles/r05_c3_adjoint_instruction_mix.txt
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
BASE = os.path.join(ROOT, "gpurun_out", "r05_ablate")
SIMDS = 1024


def short(name):
  return name.replace("void qhbm::", "").replace("(anonymous namespace)::", "").split("(")[0]


def read(variant):
  dur, ctr = {}, {}
  for f in glob.glob(f"{BASE}/{variant}/c/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
      if short(r["Kernel_Name"]).startswith("pass_adjx"):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
  for f in glob.glob(f"{BASE}/{variant}/c/**/t_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
      if short(r["Kernel_Name"]).startswith("pass_adjx") and r["Dispatch_Id"] in dur:
        ctr[r["Counter_Name"]] = ctr.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
  if not dur:
    return None
  steps = 2.0  # scripts/experiments/one_step.py runs two steps
  cycles = ctr.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / steps
  return {"ms": sum(dur.values()) / 1e6 / steps, "cycles": cycles, "valu": ctr.get("SQ_INSTS_VALU", 0.0) / steps,
          "active": 4.0 * ctr.get("SQ_ACTIVE_INST_VALU", 0.0) / steps, "salu": ctr.get("SQ_INSTS_SALU", 0.0) / steps,
          "ghz": ctr.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / max(sum(dur.values()), 1)}


def census(states):
  import bench
  from qhbmlib_amd import _engine
  gates, n_params = bench.hea_gates(20, 16)
  eng = _engine.Engine(None)
  eng.set_circuit(20, gates, n_params)
  eng.set_observables([bench.xxz_op(20)])
  tot = {}
  for row in eng.op_census(adjoint=True):
    for k, v in row.items():
      tot[k] = tot.get(k, 0.0) + v * states
  return tot


def main():
  states = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
  head = read("head") or read("base")
  if head is None:
    sys.exit("no gpurun_out/r05_ablate/head: run scripts/r05_ablate.sh on the GPU box first")
  c = census(states)
  print(f"# Config 3 (20 qubits, depth 16, XXZ), adjoint sweep of ONE step over {states} states: 9 launches of pass_adjx_kernel<12>")
  print(f"# head: {head['ms']:.2f} ms, {head['cycles'] / 1e6:.1f} M shader cycles at {head['ghz']:.3f} GHz (GRBM_GUI_ACTIVE / 8 / time), "
        f"{head['valu'] / 1e9:.2f} G VALU wave-instructions, {head['salu'] / 1e9:.2f} G SALU;")
  print(f"#       VALU active {head['active'] / (head['cycles'] * SIMDS):.3f} of the SIMD-cycles; {head['cycles'] * SIMDS / head['valu']:.2f} SIMD-cycles per VALU instruction")
  print("# piece = head - (kernel with the piece compiled out); executions = wave-executions of the micro-op (qhbm_op_census)")
  print(f"# {'piece':34s} {'VALU G':>8s} {'share':>6s} {'Mcycles':>8s} {'share':>6s} {'cyc/inst':>8s} {'executions':>11s} {'inst/exec':>9s} {'packed ops/exec (formulation)':>30s}")
  pieces = [
      ("X**t un-applied on psi and lambda", "no_x_at_all", "no_x_inner", c["x"] + c["x_no_slot"], "48 (2 x 24: three shears x 8 pairs)"),
      ("X inner products Im<lam|X|psi>", "no_x_inner", None, c["x"], "16 + 2 (sum of two partials, x - y)"),
      ("FULL tables + their ten partials", "no_full", None, c["full"], "60 (2 x 15 x 2) + 31 scalar mul/fma + 27 scalar adds"),
      ("PH1 / PH2 phases + their sums", "no_ph1_ph2", None, c["ph1"] + c["ph2"], "PH1 8 + 32, PH2 4 + 16"),
      ("boundary phases (CPH) + their reduction", "no_cph", None, c["cph_tile_on"] + c["cph_wave_on"] + c["cph_lane"], "8 + 32 per execution with the predicate on"),
      ("cross-lane reductions (add_slots8)", "no_butterfly", None, c["reduce8"], "0 (values + pairs DPP adds + 7)"),
      ("LDS exchange between rounds", "no_exchange", None, c["rounds"], "0 (64 DS ops + 60 v_xor)"),
  ]
  seen_valu = seen_cyc = 0.0
  for label, variant, minus, execs, formulation in pieces:
    v = read(variant)
    if v is None:
      continue
    ref = read(minus) if minus else head
    d_valu, d_cyc = ref["valu"] - v["valu"], ref["cycles"] - v["cycles"]
    seen_valu += d_valu
    seen_cyc += d_cyc
    print(f"  {label:34s} {d_valu / 1e9:8.3f} {d_valu / head['valu']:6.1%} {d_cyc / 1e6:8.2f} {d_cyc / head['cycles']:6.1%} "
          f"{d_cyc * SIMDS / max(d_valu, 1):8.2f} {execs:11.0f} {d_valu / max(execs, 1):9.1f} {formulation:>30s}")
  skel = read("no_instances")
  if skel:
    print(f"  {'skeleton: tile I/O, exchange, decode':34s} {skel['valu'] / 1e9:8.3f} {skel['valu'] / head['valu']:6.1%} {skel['cycles'] / 1e6:8.2f} "
          f"{skel['cycles'] / head['cycles']:6.1%}   (the kernel with NO instance executed: runs alone in this many cycles)")
  print(f"  {'sum of the pieces above':34s} {seen_valu / 1e9:8.3f} {seen_valu / head['valu']:6.1%} {seen_cyc / 1e6:8.2f} {seen_cyc / head['cycles']:6.1%}")
  noio = read("adj_no_io")
  if noio:
    print(f"# tile loads / stores compiled out (constants in, nothing stored): {noio['cycles'] / 1e6:.1f} M cycles "
          f"({1 - noio['cycles'] / head['cycles']:.1%} of the sweep's cycles are tile I/O the arithmetic does not hide; "
          f"its wall time {noio['ms']:.2f} ms also gains the clock zero data returns: {noio['ghz']:.3f} GHz)")
  packed = 48 * (c["x"] + c["x_no_slot"]) + 18 * c["x"] + 60 * c["full"] + 40 * c["ph1"] + 20 * c["ph2"] + 40 * (
      c["cph_tile_on"] + c["cph_wave_on"] + c["cph_lane"])
  print(f"# packed-fp32 operations the formulation needs (census x counts above, FULL partials as 29 packed-equivalents): "
        f"{(packed + 29 * c['full']) / 1e9:.2f} G = {(packed + 29 * c['full']) / head['valu']:.1%} of the VALU instructions issued")


if __name__ == "__main__":
  main()
