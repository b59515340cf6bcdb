"""Per-kernel table of one round: steady-state kernel times (tools/prof_summary.py output) joined with the PMC HBM traffic per launch
(tools/pmc_summary.py --json) and the matrix-pipe utilisation (tools/pmc_kernel.py --mfma) -> markdown.
usage: python tools/roofline_table.py profiles/r06z [min_ms]"""
import json, os, re, sys
pre = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
pmc = json.load(open(pre + "_pmc_traffic.json"))
busy = {}
if os.path.exists(pre + "_pmc_mfma.txt"):
    for l in open(pre + "_pmc_mfma.txt"):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+\d+\s+\d+\s+\d+\s+\d+\s+\d+\s+(.*)$", l)
        if m:
            busy[m[3].strip()] = float(m[2])
rows, total, nl = [], 0.0, 0.0
for l in open(pre + "_steady_state.txt"):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)$", l)
    if not m:
        continue
    ms, pct, calls, avg, name = float(m[1]), float(m[2]), float(m[3]), float(m[4]), m[5].strip()
    total += ms
    nl += calls
    key = next((k for k in pmc if not k.startswith("_") and (name.startswith(k) or k.startswith(name[:40]))), None)
    mb = pmc[key]["hbm_bytes_per_launch"] / 1e6 if key else None
    b = next((v for k, v in busy.items() if name.startswith(k) or k.startswith(name[:40])), None)
    rows.append((ms, name[:52], calls, avg, mb, b))
print(f"kernel time {total:.2f} ms / step over {nl:.0f} launches (rows below: >= {min_ms} ms / step)\n")
print("| kernel | ms/step | launches | avg us | HBM MB/launch (PMC) | TB/s | matrix pipe busy |")
print("|---|---|---|---|---|---|---|")
for ms, name, calls, avg, mb, b in sorted(rows, reverse=True):
    if ms < min_ms:
        continue
    print(f"| `{name}` | {ms:.2f} | {calls:.0f} | {avg:.1f} | {'%.0f' % mb if mb is not None else '-'} | "
          f"{'%.2f' % (mb / avg) if mb is not None else '-'} | {'%.2f' % b if b is not None and b > 0 else '-'} |")
