# Which phase bounds the token-Linear GEMM?  Builds copies of the library whose GEMM kernel lacks its epilogue stores (1), its
# MFMAs (2), its global operand loads (4) or combinations, and times every shape with each (results are wrong by design).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd; C=$P/csrc
D=$R/gpurun_out/diag; mkdir -p $D
others=$(ls $C/build/*.o | grep -v linear_gemm.o)
for a in ${ABLS:-0 1 2 4 3 6}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDHZ_GEMM_ABL=$a -I$R/include -I$C -c $C/linear_gemm.hip -o $D/linear_gemm_abl$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdehaze_abl$a.so $others $D/linear_gemm_abl$a.o
  echo "== DHZ_GEMM_ABL=$a"; DHZ_LIB_PATH=$D/libdehaze_abl$a.so python $R/tools/bench_gemm.py nolib
done
