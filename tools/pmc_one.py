"""FETCH_SIZE / WRITE_SIZE per launch of kernels matching a name (one rocprofv3 --pmc counter_collection.csv)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault((r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, c in by.items():
    print(d, " ".join(f"{k}={v * 1024 / 1e6:.1f}MB{'(x2=' + format(2 * v * 1024 / 1e6, '.1f') + ')' if k == 'FETCH_SIZE' else ''}" for k, v in c.items()))
