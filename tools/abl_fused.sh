# timing diagnostics: a DIAGNOSTIC copy of the library built with -DDHZ_DIAG honours the DHZ_FUSED_ABLATE bit mask, which
# skips phases of the fused attention kernel (outputs are then wrong).  The product library ignores the variable.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd/csrc
mkdir -p $R/gpurun_out/diag
for f in $C/*.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDHZ_DIAG -I$R/include -I$C -c $f -o $R/gpurun_out/diag/$(basename ${f%.hip}).o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_out/diag/libdehaze_hip_diag.so $R/gpurun_out/diag/*.o
for a in ${ABLS:-0 127}; do echo -n "abl=$a: "; DHZ_LIB_PATH=$R/gpurun_out/diag/libdehaze_hip_diag.so DHZ_FUSED_ABLATE=$a python $R/tools/bench_fused.py 2>&1 | grep -E "res  128 C   32 shift 4|res  128 C   64 shift 4|res   32 C  128 shift 4" | awk '{printf "C%s train %s eval %s | ", $4, $9, $16}'; echo; done
