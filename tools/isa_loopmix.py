"""Instruction mix of the hottest loop (the one with most matrix instructions) of every kernel of a csrc file - the audit behind the
round-5 finding that vector instructions and matrix instructions of a SIMD serialise (tools/ubench/overlap.hip, interleave.hip):
what counts in a matrix-bound loop is the NUMBER of vector instructions (any wave), so this prints it next to the MFMA count, with
the markers of integer divisions (v_rcp_iflag / v_mul_hi_u32: ~40 - 80 vector instructions each) and 64-bit address arithmetic.

    python tools/isa_loopmix.py csrc-file.hip [kernel-name-substring ...] [-DVARIANT=1 ...]
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd", "csrc")


def compile_to_asm(src, defs):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}", f"-I{CSRC}/build", "-S",
           "--cuda-device-only", "-Wno-inline-asm", src, "-o", out] + defs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    return out


def blocks_of(body):
    blocks, cur = [], None
    for l in body.splitlines():
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', l)
        if m:
            cur = {'label': m.group(1), 'comment': m.group(2) or '', 'ins': []}
            blocks.append(cur)
            continue
        if re.match(r'^; %bb\.\d+:', l):
            cur = {'label': 'bb', 'comment': l.split(':', 1)[1], 'ins': []}
            blocks.append(cur)
            continue
        if cur is None:
            continue
        t = l.split(';')[0].strip()
        if t and not t.startswith('.'):
            cur['ins'].append(t)
    return blocks


def hottest_loop(blocks):
    best, bc = None, -1
    for h in blocks:
        if 'Inner Loop Header' not in h['comment']:       # innermost loops only: a persistent tile loop would blend in its per-tile overhead
            continue
        hl = h['label'].replace('.L', '')
        mem = [h] + [b for b in blocks if re.search(r'Header=%s\b' % hl, b['comment'])]
        n = sum(1 for b in mem for i in b['ins'] if i.startswith('v_mfma'))
        if n > bc:
            bc, best = n, mem
    return best


def main(argv):
    defs = [a for a in argv if a.startswith('-D')]
    args = [a for a in argv if not a.startswith('-')]
    src = args.pop(0)
    asm = open(compile_to_asm(src if os.path.isabs(src) else os.path.join(CSRC, src), defs)).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', asm, re.M | re.S):
        name = m.group(1)
        if args and not any(a in name for a in args):
            continue
        loop = hottest_loop(blocks_of(m.group(2)))
        if not loop:
            continue
        c = Counter(i.split()[0] for b in loop for i in b['ins'])
        mf = sum(v for k, v in c.items() if k.startswith('v_mfma'))
        if not mf:
            continue
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        pk = sum(v for k, v in c.items() if k.startswith('v_pk_'))
        div = c.get('v_rcp_iflag_f32_e32', 0) + c.get('v_rcp_f32_e32', 0)
        a64 = c.get('v_lshl_add_u64', 0) + c.get('v_mad_u64_u32', 0) + c.get('v_addc_co_u32_e32', 0)
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace('void (anonymous namespace)::', '').split('(')[0]
        print(f"{dem[:60]:60s} MFMA {mf:4d}  vector {valu:4d} ({valu / mf:4.1f} per MFMA; packed-fp32 {pk:3d})  reciprocals {div:2d}  64-bit address ops {a64:3d}  "
              f"LDS {sum(v for k, v in c.items() if k.startswith('ds_')):3d}  VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))):3d}")
        if '-v' in argv:
            for k, v in c.most_common(40):
                print(f"      {v:5d} {k}")


if __name__ == "__main__":
    main(sys.argv[1:])
