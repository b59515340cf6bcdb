"""Per-kernel roofline table of one round: steady-state kernel times (tools/prof_summary.py output) joined with the PMC HBM
traffic per launch (tools/pmc_summary.py --json) -> markdown.  usage: python tools/roofline_table.py profiles/r01p"""
import json, re, sys
pre = sys.argv[1]
pmc = json.load(open(pre + "_pmc_traffic.json"))
rows, total = [], 0.0
for l in open(pre + "_steady_state.txt"):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)$", l)
    if not m:
        continue
    ms, pct, calls, avg, name = float(m[1]), float(m[2]), float(m[3]), float(m[4]), m[5].strip()
    total += ms
    key = next((k for k in pmc if name.startswith(k) or k.startswith(name[:40])), None)
    if key is None:
        continue
    mb = pmc[key]["hbm_bytes_per_launch"] / 1e6
    rows.append((ms, name[:48], calls, avg, mb, mb / avg))                 # MB per us = TB/s
print("| kernel | ms/step | launches/step | avg µs | HBM MB/launch (PMC) | TB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for ms, name, calls, avg, mb, tbs in sorted(rows, reverse=True):
    print(f"| `{name}` | {ms:.2f} | {calls:.0f} | {avg:.1f} | {mb:.0f} | {tbs:.2f} | {tbs / 8:.2f} |")
