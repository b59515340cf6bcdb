"""Per-kernel means of rocprofv3 --pmc counter CSVs.
  python tools/pmc_kernel.py <dir> <kernel substring>            raw counters
  python tools/pmc_kernel.py <dir> --mfma [regex]                 matrix-pipe utilisation table (needs the counters of
                                                                  tools/profile_round.sh's MFMA pass)
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the fraction of all SIMD-cycles of the
dispatch in which a matrix instruction occupied the pipe (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs,
MI355X_MICROARCH.md 'DVFS give-back').  VALU/wave counts every vector instruction a wave issued (matrix ones included)."""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:70]


if len(sys.argv) > 2 and sys.argv[2] == "--mfma":
    pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else ".")
    rows = []
    for k, d in acc.items():
        if not pat.search(k) or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
            continue
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        n = len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
        util = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024) if gui else float("nan")
        waves = mean.get("SQ_WAVES", 0.0)
        rows.append((mean["SQ_VALU_MFMA_BUSY_CYCLES"] * n, short(k), n, util, mean.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / 4,
                     mean.get("SQ_INSTS_VALU", 0.0) / waves if waves else 0.0,
                     4 * mean.get("SQ_WAIT_INST_ANY", 0.0) / waves if waves else 0.0,
                     4 * mean.get("SQ_WAIT_ANY", 0.0) / waves if waves else 0.0,
                     4 * mean.get("SQ_WAVE_CYCLES", 0.0) / waves if waves else 0.0))
    print(f"# {'launches':>8} {'MFMA util':>9} {'MFMA instr':>12} {'VALU/wave':>10} {'issue-stall cyc/wave':>21} {'waitcnt cyc/wave':>17} {'cycles/wave':>12}  kernel")
    for _, k, n, util, mf, vw, wi, wa, wc in sorted(rows, reverse=True):
        print(f"  {n:8d} {util:9.3f} {mf:12.0f} {vw:10.0f} {wi:21.0f} {wa:17.0f} {wc:12.0f}  {k}")
else:
    for k, d in acc.items():
        if sys.argv[2] in k:
            print(short(k))
            for c, v in sorted(d.items()):
                print(f"   {c:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
