"""Per-kernel difference of two steady-state tables (tools/prof_summary.py output): which kernels a change moved, in ms per step.
usage: python tools/prof_diff.py profiles/r02h_steady_state.txt gpurun_out/<new>_steady_state.txt [--top 12]"""
import sys


def load(f):
    d = {}
    for l in open(f):
        if l.startswith('#'):
            continue
        p = l.split(None, 4)
        if len(p) == 5:
            d[p[4].strip()] = float(p[0])
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
rows = sorted(((b.get(k, 0.0) - a.get(k, 0.0), k) for k in set(a) | set(b)), reverse=True)
print(f"total {sum(a.values()):.3f} -> {sum(b.values()):.3f} ms/step (kernels listed in both tables' --top ranges only)")
for dlt, k in rows[:top] + [(None, "...")] + rows[-top:]:
    print("   ..." if dlt is None else f"{dlt:+8.3f}  {a.get(k, 0.0):7.3f} -> {b.get(k, 0.0):7.3f}  {k[:90]}")
