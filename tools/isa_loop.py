"""Skeleton of one kernel's ISA: waits, barriers, memory and matrix instructions in order (runs of the same opcode collapsed).
usage: python tools/isa_loop.py file.s <substring of the mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(_Z\w+):', s, re.M) if sys.argv[2] in m.group(1)]
name = names[0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].splitlines()
print(name, len(body), 'lines')
keep = re.compile(r's_waitcnt|s_barrier|global_load|global_store|global_atomic|ds_read|ds_write|v_mfma|s_cbranch|^\.LBB|buffer_|scratch_')
out = []; last = None; cnt = 0
for l in body:
    l = l.strip()
    if not keep.search(l): continue
    key = l.split()[0] if not l.startswith('.LBB') else l
    if key == last and not l.startswith('s_waitcnt'):
        cnt += 1; continue
    if cnt: out.append(f'      ... x{cnt+1}'); cnt = 0
    out.append(l[:120]); last = key
if cnt: out.append(f'      ... x{cnt+1}')
print('\n'.join(out))
