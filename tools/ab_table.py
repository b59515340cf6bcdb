"""Per-shape table of an A/B run of tools/bench_gemm.py made with tools/ab.sh:  python tools/ab_table.py gpurun_out/<tag>.txt"""
import re, sys
txt = open(sys.argv[1]).read()
parts = re.split(r'== (\S+) \(pass (\d)\)\n', txt)
data = {}
for i in range(1, len(parts), 3):
    rows = []
    for l in parts[i + 2].split('\n'):
        m = re.match(r'\s*(\d+)\s+(\d+)\s+(\d+) \|\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+)', l)
        if m:
            rows.append([float(x) for x in m.groups()])
    data.setdefault(parts[i], []).append(rows)
keys = list(data)
print("      T     K     N | fwd " + " ".join(f"{k:>7}" for k in keys) + " | dgrad " + " ".join(f"{k:>7}" for k in keys) + " | floor us (157 TF / 6.3 TB/s)")
n = len(data[keys[0]][0])
for j in range(n):
    r = data[keys[0]][0][j]
    T, K, N = r[0], r[1], r[2]
    fl = max(2 * T * K * N / 157.3e12, T * (K + N) * 4 / 6.3e12) * 1e6
    f = [min(p[j][4] for p in data[k]) for k in keys]
    d = [min(p[j][8] for p in data[k]) for k in keys]
    print(f"{int(T):7d} {int(K):5d} {int(N):5d} |     " + " ".join(f"{x:7.1f}" for x in f) + " |       " + " ".join(f"{x:7.1f}" for x in d) + f" | {fl:7.1f}")
