"""CPU suite: the generated code of csrc/split6_gemm.hip keeps its inline-assembly loads (gload32) and their counted waits (wait_regs)
apart the way the source assumes - no copy, spill or early use of a destination register between such a load and the first vmcnt wait
behind it, no scratch memory (tools/isa_check.py; round-4 advice).  hipcc cross-compiles here; ~10 s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_split6_inline_asm_loads_are_not_touched_before_their_wait(capsys):
    import isa_check
    path = isa_check.compile_to_asm(os.path.join(isa_check.CSRC, "split6_gemm.hip"))
    try:
        rc = isa_check.main([path])
        out = capsys.readouterr().out
        assert rc == 0, out
        assert out.count("ok  ") >= 7 and "FAIL" not in out and "SCRATCH" not in out, out
        # the checker sees what it is meant to see: plant a copy of a destination register right behind one load pair
        lines = open(path).read().splitlines()
        k0 = next(i for i, ln in enumerate(lines) if ln.startswith("_Z") and "split6_wide_kernel" in ln)      # inside a checked kernel
        k = next(i for i, ln in enumerate(lines) if i > k0 and "global_load_dwordx4" in ln and "offset:16" in ln)
        import re
        dst = re.search(r"global_load_dwordx4\s+v\[(\d+):", lines[k]).group(1)
        lines.insert(k + 1, f"\tv_mov_b32_e32 v255, v{dst}")
        bad = path + ".planted.s"
        open(bad, "w").write("\n".join(lines) + "\n")
        assert isa_check.main([bad]) == 1
        assert "FAIL" in capsys.readouterr().out
        os.unlink(bad)
    finally:
        os.unlink(path)


def test_winograd43_dma_requests_keep_their_base_in_sgprs_and_own_m0():
    """csrc/winograd43_conv.hip issues its LDS-DMA requests from inline assembly (s_mov_b32 m0 / global_load_lds_dwordx4 v_off, s[base]): the
    compiler is told that m0 is clobbered but treats it as a reserved register, so the kernel must not contain any compiler-generated use of
    m0 (none of its own LDS-DMA builtins, no movrel / GWS / sendmsg), every request must be the SGPR-base form (no 64-bit vector address
    arithmetic between MFMAs), and the 512-register kernel must not touch scratch memory."""
    import re
    import isa_check
    path = isa_check.compile_to_asm(os.path.join(isa_check.CSRC, "winograd43_conv.hip"))
    try:
        asm = open(path).read()
        kernels = re.findall(r"^(_Z\w*winograd43_conv3x3_kernel\w*):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.M | re.S)
        assert len(kernels) == 2
        for name, body in kernels:
            inside, outside_m0, dma, dma_sgpr = False, 0, 0, 0
            for ln in body.splitlines():
                t = ln.split(";")[0] if not ln.lstrip().startswith(";;#") else ln
                if "#ASMSTART" in ln:
                    inside = True
                elif "#ASMEND" in ln:
                    inside = False
                elif re.search(r"\bm0\b", t) and not inside:
                    outside_m0 += 1
                if "global_load_lds_dwordx4" in t:
                    dma += 1
                    dma_sgpr += bool(re.search(r"global_load_lds_dwordx4\s+v\d+,\s*s\[\d+:\d+\]", t)) and inside
            assert outside_m0 == 0, (name, outside_m0)
            assert dma >= 40 and dma == dma_sgpr, (name, dma, dma_sgpr)
            assert not re.search(r"^\s*scratch_(load|store)", body, re.M), name
            import isa_loopmix
            loop = isa_loopmix.hottest_loop(isa_loopmix.blocks_of(body))
            ops_ = [i.split()[0] for b in loop for i in b["ins"]]
            assert ops_.count("v_mfma_f32_16x16x4_f32") == 144, name
            assert not any(o in ("v_lshl_add_u64", "v_mad_u64_u32", "v_add_co_u32_e32") for o in ops_), name      # no vector address arithmetic in the channel loop
    finally:
        os.unlink(path)


def test_winograd43_accumulator_hazards_are_covered_by_the_code_shape():
    """The 288 accumulators of csrc/winograd43_conv.hip are written by inline-asm MFMAs the compiler's hazard recogniser cannot see
    (ADVICE round 5).  Per code object: (1) inside the channel loop no instruction other than those MFMAs reads or writes an
    accumulator register (no v_mov / v_accvgpr copy or live-range split next to a matrix instruction that is still in flight);
    (2) the channel loop ENDS with the sixteen matrix instructions of the VGPR-class accumulators, so an AGPR-class accumulator is
    at least 16 matrix instructions old when the loop is left, and on every path from the loop's exits two `s_nop 15` come before
    the first instruction that touches a VGPR-class accumulator (the fence carries them as operands)."""
    import re
    import isa_check
    import isa_loopmix
    areg = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")

    def regs(text):
        out = set()
        for m in areg.finditer(text):
            if m.group(1):
                out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
            else:
                out.add((m.group(4), int(m.group(5))))
        return out

    path = isa_check.compile_to_asm(os.path.join(isa_check.CSRC, "winograd43_conv.hip"))
    try:
        asm = open(path).read()
        kernels = re.findall(r"^(_Z\w*winograd43_conv3x3_kernel\w*):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.M | re.S)
        assert len(kernels) == 2
        for name, body in kernels:
            loop = [i for b in isa_loopmix.hottest_loop(isa_loopmix.blocks_of(body)) for i in b["ins"]]
            mf = [i for i in loop if i.startswith("v_mfma_f32_16x16x4_f32")]
            assert len(mf) == 144, name
            accs = set()
            for i in mf:
                accs |= regs(i.split(None, 1)[1].split(",")[0])
            assert len(accs) == 288 and sum(1 for c, _ in accs if c == "a") == 256, (name, len(accs))
            vaccs = {r for r in accs if r[0] == "v"}
            for i in mf[-16:]:                         # the loop's tail: the VGPR-class positions 32 .. 35
                assert regs(i.split(None, 1)[1].split(",")[0]) <= vaccs, (name, i)
            # (1) nothing but the matrix instructions touches an accumulator inside the loop
            for i in loop:
                if i.startswith("v_mfma") or " " not in i:
                    continue
                hit = regs(i.split(None, 1)[1]) & accs
                assert not hit, (name, i, sorted(hit)[:4])
            # (2) control flow: from the exits of the channel loop, every path reaches the `s_nop 15` pair before it reaches an instruction
            #     that touches a VGPR-class accumulator register (block layout is not program order: walk the labels and branches)
            blocks = isa_loopmix.blocks_of(body)
            label_at = {b["label"]: k for k, b in enumerate(blocks) if b["label"].startswith(".LBB")}
            hdr = next(b for b in blocks if any(i.startswith("v_mfma") for i in b["ins"]) and "Inner Loop Header" in b["comment"])
            hl = hdr["label"].replace(".L", "")
            in_loop = {k for k, b in enumerate(blocks) if b is hdr or re.search(r"Header=%s\b" % hl, b["comment"])}

            def succ(k):
                out, ins = [], blocks[k]["ins"]
                for t in ins:
                    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
                    if m:
                        out.append(label_at[m.group(1) or m.group(2)])
                if not (ins and (ins[-1].startswith("s_branch") or ins[-1].startswith("s_endpgm"))) and k + 1 < len(blocks):
                    out.append(k + 1)
                return out

            def scan(k):
                """-> 'nops' (the pair comes first), 'acc' (an accumulator access comes first) or None (neither in this block)"""
                n = 0
                for t in blocks[k]["ins"]:
                    if t.startswith("s_nop") and t.split()[1] == "15":
                        n += 1
                        if n == 2:
                            return "nops"
                    elif " " in t and not t.startswith("v_mfma") and regs(t.split(None, 1)[1]) & vaccs:
                        return "acc"
                return None

            todo = [t for k in in_loop for t in succ(k) if t not in in_loop]
            assert todo, name
            seen, protected = set(), 0
            while todo:
                k = todo.pop()
                if k in seen:
                    continue
                seen.add(k)
                r = scan(k)
                assert r != "acc", (name, blocks[k]["label"], "an accumulator is touched before the s_nop pair")
                if r == "nops":
                    protected += 1
                    continue
                todo += [t for t in succ(k) if t not in in_loop]
            assert protected >= 1, name
    finally:
        os.unlink(path)
