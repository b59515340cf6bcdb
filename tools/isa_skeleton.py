"""Control-flow / wait skeleton of one kernel in a hipcc -S listing:  python tools/isa_skeleton.py file.s <mangled-name substring> [--full]
Prints labels, branches, barriers, s_waitcnt vmcnt and the first store / load of every run - enough to see where the compiler
drains the vector-memory counter (store-vmcnt(0)-store sequences, loop-preheader flushes)."""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(s) if re.match(r'^_Z\S*:', l) and key in l][0]
end = [i for i in range(start, len(s)) if s[i].strip().startswith('s_endpgm')][0]
body = s[start:end]
last = None
n_mfma = 0
for i, l in enumerate(body):
    t = l.strip()
    kind = None
    if re.match(r'(s_waitcnt vmcnt|s_waitcnt lgkmcnt\(0\)|s_barrier|s_cbranch|s_branch)', t) or t.startswith('.LBB'):
        kind = 'ctl'
    elif t.startswith('global_store') or t.startswith('global_atomic'):
        kind = 'st'
    elif t.startswith('global_load') or t.startswith('buffer_load'):
        kind = 'ld'
    elif t.startswith('ds_write') or t.startswith('ds_store'):
        kind = 'dsw'
    elif t.startswith('ds_read') or t.startswith('ds_load'):
        kind = 'dsr'
    elif t.startswith('v_mfma'):
        kind = 'mfma'
    if kind is None:
        continue
    if kind == 'ctl':
        if 'lgkmcnt(0)' in t and '--full' not in sys.argv:
            continue
        print(f"{i:6d}  {t[:80]}")
        last = None
    elif kind != last:
        print(f"{i:6d}      [{kind} ...]")
        last = kind
for l in s[end:end + 60]:
    if re.search(r'NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize', l):
        print(l.strip())
