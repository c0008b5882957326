"""LayerNorm + LeakyReLU of one MLP block [16000, 512]: stock layers vs the fused HIP pass, forward and forward+backward."""
import sys, time, torch, torch.nn as nn
sys.path.insert(0, '.')
from ddsp_pytorch_amd.decoder import _LayerNormLeakyReLU
def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
x = torch.randn(32, 500, 512, device='cuda', requires_grad=True)
ln = nn.LayerNorm(512).cuda(); act = nn.LeakyReLU()
w = torch.randn(32, 500, 512, device='cuda')
def stock(): return act(ln(x))
def fused(): return _LayerNormLeakyReLU.apply(x, ln.weight, ln.bias, ln.eps, act.negative_slope)
for name, f in (("stock", stock), ("fused", fused)):
    with torch.no_grad():
        fw = timeit(f)
    def fb():
        y = f(); (y * w).sum().backward()
    print({"impl": name, "fwd_ms": round(fw, 4), "fwd_bwd_ms(incl. mul+sum)": round(timeit(fb), 4)}, flush=True)
