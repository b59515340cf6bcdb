"""Join a rocprofv3 run made with --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE [...]: per kernel name + grid the mean
duration, the effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the matrix-pipe utilisation.
    python tools/pmc_clock.py <rocprof output dir> [regex]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
trace = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        trace[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size_X"], r["Workgroup_Size_X"])
cnt = collections.defaultdict(dict)
names = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
agg = collections.defaultdict(list)
for did, c in cnt.items():
    if did not in trace or not pat.search(names[did]):
        continue
    dur, gx, wx = trace[did]
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", names[did]).split("(")[0][:60]
    agg[(n, int(gx) // int(wx))].append((dur, c.get("GRBM_GUI_ACTIVE", 0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
print(f"# {'n':>4} {'us':>8} {'GHz':>6} {'MFMA busy':>9}  grid  kernel")
for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    dur = sum(x[0] for x in v) / len(v)
    gui = sum(x[1] for x in v) / len(v)
    mf = sum(x[2] for x in v) / len(v)
    print(f"  {len(v):4d} {dur / 1e3:8.1f} {gui / 8 / dur:6.2f} {mf / (gui / 8 * 1024) if gui else 0:9.3f}  {g:5d}  {n}")
