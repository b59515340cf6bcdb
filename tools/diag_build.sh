# builds the DIAGNOSTIC copy of the library (-DDHZ_DIAG: honours the DHZ_* tuning / ablation environment switches the product
# library compiles out) into gpurun_out/diag/libdehaze_hip_diag.so; use it with DHZ_LIB_PATH=...
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd/csrc
mkdir -p $R/gpurun_out/diag
for f in $C/*.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDHZ_DIAG -I$R/include -I$C -c $f -o $R/gpurun_out/diag/$(basename ${f%.hip}).o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_out/diag/libdehaze_hip_diag.so $R/gpurun_out/diag/*.o
echo $R/gpurun_out/diag/libdehaze_hip_diag.so
