# A/B of compile-time variants of ONE kernel file, same box, alternating passes:
#   SRC=linear_gemm VARIANTS="-DDHZ_GEMM_NT=0 -DDHZ_GEMM_NT=1" CMD="python tools/bench_gemm.py nolib" bash tools/variants.sh
# Each variant is linked into its own copy of the library (gpurun_out/diag/libdehaze_v<i>.so) and CMD runs with DHZ_LIB_PATH set.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd; C=$P/csrc
D=$R/gpurun_out/diag; mkdir -p $D
SRC=${SRC:-linear_gemm}
[ -f $C/build/api.o ] || bash $C/build.sh > /dev/null
others=$(ls $C/build/*.o | grep -v "/$SRC.o" | grep -v "/api.o")
i=0
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${v//,/ } -I$R/include -I$C -c $C/$SRC.hip -o $D/${SRC}_v$i.o
  # api.o per variant: dhz_build_id() of a variant library carries the variant's tag, so bench.py never pairs it with the product's PMC table
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDHZ_VARIANT_TAG=\"${SRC}_v$i\" -I$R/include -I$C -I$C/build -c $C/api.hip -o $D/api_v$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdehaze_v$i.so $others $D/api_v$i.o $D/${SRC}_v$i.o
  i=$((i+1))
done
for rep in $(seq 1 ${PASSES:-2}); do i=0; for v in $VARIANTS; do echo "== $v (pass $rep)"; DHZ_LIB_PATH=$D/libdehaze_v$i.so $CMD; i=$((i+1)); done; done
