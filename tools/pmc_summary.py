"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel.
FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced streaming reads, i.e. exactly half
the bytes (MI355X_MICROARCH.md, HBM section) -> doubled here; WRITE_SIZE is exact.  Units in the CSV: KiB.
usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [--json out.json]
"""
import argparse
import collections
import csv
import json
import re


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]].append(float(r["Counter_Value"]))
    return agg


ap = argparse.ArgumentParser()
ap.add_argument("fetch")
ap.add_argument("write")
ap.add_argument("--json")
ap.add_argument("--stamp", action="store_true", help="record the loaded library's dhz_build_id() in the JSON (bench.py checks it)")
ap.add_argument("--filter", default="ps_attn|linear_wgrad|leff|ln_partition|reverse_residual|charbonnier|adamw|bias_")
a = ap.parse_args()
f, w = load(a.fetch, "FETCH_SIZE"), load(a.write, "WRITE_SIZE")
out = {}
print(f"# {'kernel':40s} {'launches':>8} {'read MB/launch (2x FETCH)':>26} {'write MB/launch':>16} {'HBM MB/launch':>14}")
for k in sorted(f):
    if not re.search(a.filter, k):
        continue
    rd = 2 * sum(f[k]) / len(f[k]) * 1024 / 1e6
    wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) * 1024 / 1e6
    print(f"  {k:40s} {len(f[k]):8d} {rd:26.2f} {wr:16.2f} {rd + wr:14.2f}")
    out[k] = {"launches": len(f[k]), "read_bytes_per_launch": rd * 1e6, "write_bytes_per_launch": wr * 1e6,
              "hbm_bytes_per_launch": (rd + wr) * 1e6}
if a.stamp:
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
    from dehaze_hip import _lib
    out["_stamp"] = {"build_id": _lib.load().dhz_build_id().decode(), "command": "bench.py --steps 3 --warmup 2 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
