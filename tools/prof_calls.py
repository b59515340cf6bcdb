"""Per-call durations of one kernel inside a steady-state step (rocprofv3 --kernel-trace CSV of bench.py).
usage: python tools/prof_calls.py <kernel_trace.csv> <kernel-name-substring> [--step 6]"""
import argparse, csv
ap = argparse.ArgumentParser()
ap.add_argument("trace"); ap.add_argument("name"); ap.add_argument("--step", type=int, default=6)
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
step, out = 0, []
for r in rows:
    if "adamw_kernel" in r["Kernel_Name"]:
        step += 1
        continue
    if step == a.step and a.name in r["Kernel_Name"]:
        out.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        print(f"{out[-1]:9.1f} us  grid {grid}  {r['Kernel_Name'][:80]}")
print(f"{len(out)} calls, total {sum(out) / 1e3:.3f} ms")
