# SQ counters of the six-term GEMM kernels on one shape: bash tools/pmc_split6.sh   (on the GPU box, from the repo root)
R=$(pwd); O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["R"], "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
T, K, N = 32768, 256, 1024
x = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev); y = torch.empty(T, N, device=dev)
pl = [torch.empty(N * K, dtype=torch.bfloat16, device=dev) for _ in range(3)]
_lib.call("dhz_split3_planes", w.data_ptr(), N * K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), s)
for _ in range(5):
    _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s)
torch.cuda.synchronize()
PY
export R
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  for tile in 1 2; do
    rm -rf $O/p_s6
    DHZ_S6_TILE=$tile rocprofv3 --pmc $set --output-format csv -d $O/p_s6 -- python3 /tmp/one_gemm.py > /dev/null 2>&1
    f=$(ls $O/p_s6/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" $tile <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "split6" in r["Kernel_Name"] and "planes" not in r["Kernel_Name"]]
acc = collections.OrderedDict(); n = collections.Counter()
for r in rows:
    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("tile", sys.argv[2], rows[0]["Kernel_Name"][:40] if rows else "-", " ".join(f"{k}={v / n[k]:.4g}" for k, v in acc.items()))
PY
  done
done
rm -rf $O/p_s6
