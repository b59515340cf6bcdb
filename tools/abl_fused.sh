# timing diagnostics: DHZ_FUSED_ABLATE bit mask skips phases of the fused kernel (outputs are then wrong)
for a in ${ABLS:-0 127}; do echo -n "abl=$a: "; DHZ_FUSED_ABLATE=$a python tools/bench_fused.py 2>&1 | grep -E "res  128 C   32 shift 4|res  128 C   64 shift 4|res   32 C  128 shift 4" | awk '{printf "C%s train %s eval %s | ", $4, $9, $16}'; echo; done
