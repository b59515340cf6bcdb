"""Per-step kernel summary from a rocprofv3 --kernel-trace CSV of bench.py.

Steps are delimited by the AdamW kernel (exactly one launch per training step); the first `--skip` steps
(warm-up: MIOpen find-mode, allocator growth) are dropped so that the table reflects steady-state steps.
usage: python tools/prof_summary.py gpurun_out/prof/*/..._kernel_trace.csv --skip 3 [--top 40]
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    if m:
        name = m.group(1)
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip", type=int, default=3)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--delim", default="adamw_kernel", help="kernel (substring) launched exactly once per step: the step delimiter (inference runs: input_proj_fwd)")
    ap.add_argument("--launches", default=None, help="regex: also list every launch of the matching kernels in ONE steady step (grid x block, us, gap to the previous kernel)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    step, kept, nsteps = 0, [], 0
    for r in rows:
        if step >= a.skip:
            kept.append(r)
        if a.delim in r["Kernel_Name"]:
            step += 1
    nsteps = step - a.skip
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in kept:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        agg[k][0] += d
        agg[k][1] += 1
    span = (int(kept[-1]["End_Timestamp"]) - int(kept[0]["Start_Timestamp"])) / 1e6
    tot = sum(v[0] for v in agg.values()) / 1e3
    print(f"# steady-state steps: {nsteps}; wall span {span / nsteps:.2f} ms/step; kernel time {tot / nsteps:.2f} ms/step")
    print(f"# {'ms/step':>8} {'%':>6} {'calls/step':>10} {'avg us':>9}  kernel")
    for k, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"  {us / 1e3 / nsteps:8.3f} {100 * us / 1e3 / tot:6.2f} {n / nsteps:10.1f} {us / n:9.1f}  {k}")
    if a.launches:
        launches(kept, a.launches)


def launches(kept, pattern):
    """every launch of the kernels matching `pattern` within the last complete step, in launch order"""
    ends = [i for i, r in enumerate(kept) if "adamw_kernel" in r["Kernel_Name"]]
    if len(ends) < 2:
        return
    lo, hi = ends[-2] + 1, ends[-1] + 1
    rx = re.compile(pattern)
    gaps = []
    print(f"# launches of /{pattern}/ in one step ({hi - lo} kernels in the step)")
    for i in range(lo, hi):
        r = kept[i]
        gap = (int(r["Start_Timestamp"]) - int(kept[i - 1]["End_Timestamp"])) / 1e3
        gaps.append(gap)
        if rx.search(r["Kernel_Name"]):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            grid = r.get("Grid_Size_X", r.get("Grid_Size", "?")); wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
            print(f"  {d:9.1f} us  gap {gap:6.1f}  grid {grid:>9} wg {wg:>4}  {short(r['Kernel_Name'])}")
    gaps.sort()
    print(f"# gaps before the {len(gaps)} kernels of the step: sum {sum(gaps):.1f} us, median {gaps[len(gaps) // 2]:.2f}, p90 {gaps[int(len(gaps) * 0.9)]:.2f}, max {gaps[-1]:.1f}")


if __name__ == "__main__":
    main()
