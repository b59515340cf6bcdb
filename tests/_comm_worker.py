"""Worker of tests/test_gpu_ddp.py::test_c_abi_comm_two_ranks_bucketed_exchange: one process per GPU, the gradient exchange of
dehaze_hip.train.GradReducer's buckets driven through the C-ABI alone (dhz_comm_* of include/dehaze_hip.h, RCCL underneath) - what a host
that binds only libdehaze_hip.so does in place of torch.distributed.   argv: rank world id_file"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch  # noqa: E402
from dehaze_hip import _lib  # noqa: E402
from dehaze_hip.train import GradReducer  # noqa: E402

rank, world, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
uid = (ctypes.c_char * 128)()
if rank == 0:
    _lib.call("dhz_comm_unique_id", ctypes.cast(uid, ctypes.c_void_p))
    with open(id_file + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        assert time.time() - t0 < 120, "rank 0 never published the unique id"
        time.sleep(0.05)
    ctypes.memmove(uid, open(id_file, "rb").read(), 128)
comm = ctypes.c_void_p()
_lib.call("dhz_comm_init", ctypes.cast(ctypes.pointer(comm), ctypes.c_void_p), rank, world, ctypes.cast(uid, ctypes.c_void_p))
# the bucket plan of the product's reducer over a parameter set with the model's size mix (no process group: world = 1 inside)
g = torch.Generator().manual_seed(3)
params = [torch.nn.Parameter(torch.randn(n, generator=g).to(dev)) for n in (3 * 2048 * 512, 225 * 16, 512, 2048 * 512, 64 * 27, 7)]
red = GradReducer(params=params, bucket_mb=4.0)
assert len(red.buckets) >= 2
red.flat.copy_(torch.arange(red.flat.numel(), device=dev, dtype=torch.float32) % 1024 * (rank + 1))
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for lo, hi, _ in red.buckets:                                  # launch order = backward order, each on the exchange stream
        _lib.call("dhz_comm_allreduce_sum_f32", comm, red.flat.data_ptr() + 4 * lo, hi - lo, st.cuda_stream)
st.synchronize()
want = torch.arange(red.flat.numel(), device=dev, dtype=torch.float32) % 1024 * sum(r + 1 for r in range(world))
assert torch.equal(red.flat, want), (red.flat - want).abs().max().item()
assert params[0].grad.data_ptr() == red.flat.data_ptr()            # the buckets ARE the parameters' gradients (no copies)
_lib.call("dhz_comm_destroy", comm)
print(f"rank {rank}: ok, {len(red.buckets)} buckets")
