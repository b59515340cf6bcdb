"""Build-time ISA check of the kernels that issue vector-memory loads from inline assembly with the wait in a LATER asm statement
(csrc/split6_gemm.hip: gload32 / wait_regs): hipcc treats the destination registers of such a load as valid at once,
so nothing but the shape of the generated code keeps it from copying or spilling them before the counted `s_waitcnt vmcnt(N)` that
makes them valid.  This script proves that shape for the code object at hand:

  * every inline-asm load pair is followed in program order to the first `s_waitcnt vmcnt(...)` behind it: an instruction in between
    that reads or writes one of its destination registers is a violation (a v_mov / v_accvgpr copy, a spill, an early use) - the
    matching counted wait is that wait or a later one;
  * no kernel of the file may use scratch memory (a spill of such a register would be exactly that).

    python tools/isa_check.py [file.s | --compile csrc-file.hip] [kernel-name-substring ...]      exit status 1 on a violation

tests/test_isa.py (CPU suite) runs it on split6_gemm.hip."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd", "csrc")

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}", f"-I{CSRC}/build", "-S",
           "--cuda-device-only", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    return out


def kernels(asm):
    """[(name, [instruction text, ...])] - labels kept as '.LBBx:' entries"""
    res = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.M | re.S):
        body = []
        for line in m.group(2).splitlines():
            line = line.split(";")[0].strip()
            if not line or (line.startswith(".") and not line.startswith(".LBB")):
                continue
            body.append(line)
        res.append((m.group(1), body))
    return res


def check_kernel(name, body):
    """-> (violations, n_asm_loads, longest window).  The guarded loads are the INLINE-ASM ones - gload32 is the one place that emits
    two consecutive global_load_dwordx4 from the same address register pair, the second with `offset:16`; loads the compiler emits
    itself are its own responsibility.  Rule (sound, not complete): from such a load to the FIRST `s_waitcnt vmcnt(...)` behind it in
    program order (fall-through across labels and conditional branches; the scan stops at an unconditional branch), no instruction
    may read or write its destination registers - the matching counted wait is that one or a later one, so the registers are
    certainly not valid before it."""
    viol, nloads, longest = [], 0, 0
    for idx, ins in enumerate(body):
        if not (ins.startswith("global_load_dwordx4") and "offset:16" not in ins and idx + 1 < len(body)
                and body[idx + 1].startswith("global_load_dwordx4") and "offset:16" in body[idx + 1]
                and ins.split(",")[1].strip() == body[idx + 1].split(",")[1].strip()):
            continue
        nloads += 1
        dst = regs_of(ins.split(None, 1)[1].split(",")[0]) | regs_of(body[idx + 1].split(None, 1)[1].split(",")[0])
        k = idx + 2
        while k < len(body):
            cur = body[k]
            op = cur.split()[0]
            if op == "s_waitcnt" and "vmcnt" in cur:
                break
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                break
            if not op.startswith(".LBB") and " " in cur:
                hit = regs_of(cur.split(None, 1)[1]) & dst
                if hit:
                    viol.append((k, cur, sorted(hit)))
            k += 1
        longest = max(longest, k - idx)
    return viol, nloads, longest


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    if "--compile" in argv:
        src = args.pop(0)
        path = compile_to_asm(src if os.path.isabs(src) else os.path.join(CSRC, src))
    else:
        path = args.pop(0) if args else compile_to_asm(os.path.join(CSRC, "split6_gemm.hip"))
    asm = open(path).read()
    wanted = args or ["split6_"]
    bad = 0
    if re.search(r"^\s*scratch_(load|store)", asm, re.M) or re.search(r"\.private_segment_fixed_size:\s*[1-9]", asm):
        for m in re.finditer(r"\.name:\s*(\S+)\s*\n(?:.*\n){0,12}?\s*\.private_segment_fixed_size:\s*([1-9]\d*)", asm):
            if any(w in m.group(1) for w in wanted):
                print(f"SCRATCH {m.group(1)}: private segment {m.group(2)} bytes")
                bad += 1
    for name, body in kernels(asm):
        if not any(w in name for w in wanted):
            continue
        viol, nloads, longest = check_kernel(name, body)
        print(f"{'FAIL' if viol else 'ok  '} {name}: {len(body)} instructions, {nloads} inline-asm load pairs, longest load-to-first-wait window "
              f"{longest} instructions, {len(viol)} violation(s)")
        if nloads == 0:
            print("      (no inline-asm load pair recognised: the pattern of gload32 changed - update tools/isa_check.py)")
            bad += 1
        for idx, ins, regs in viol[:8]:
            print(f"      [{idx}] {ins}    <- in-flight destination registers v{regs}")
        bad += len(viol)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
