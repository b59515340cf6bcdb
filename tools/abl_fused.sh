# timing diagnostics: variant copies of the library with phases of the fused attention kernel compiled OUT
# (-DDHZ_FUSED_ABL=<bit mask>, outputs are then wrong); the product library has no such switch.   ABLS="0 1 2 127" bash tools/abl_fused.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
V=""; for a in ${ABLS:-0 127}; do V="$V -DDHZ_FUSED_ABL=$a"; done
SRC=fused_attn VARIANTS="$V" PASSES=1 CMD="python $R/tools/bench_fused.py" bash $R/tools/variants.sh 2>&1 | grep -E "^==|res  128 C   32 shift 4|res  128 C   64 shift 4|res   32 C  128 shift 4"
