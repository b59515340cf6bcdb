"""OutputProj (Conv2d 64 -> 3, 3x3) on the library: does padding the output channels pick a better kernel?"""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(32, 128 * 128, 64, device=dev).view(32, 128, 128, 64).permute(0, 3, 1, 2).requires_grad_()
for co in (3, 4, 8, 16, 32):
    w = torch.randn(co, 64, 3, 3, device=dev, requires_grad=True)
    g = torch.randn(32, co, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
    tf = timeit(lambda: F.conv2d(x, w, None, 1, 1))
    y = F.conv2d(x, w, None, 1, 1)
    def bwd():
        return torch.autograd.grad(y, (x, w), g, retain_graph=True)
    tb = timeit(bwd)
    print(f"Cout {co:3d}: fwd {tf:7.1f} us   bwd (dgrad + wrw) {tb:7.1f} us")
