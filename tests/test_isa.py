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
