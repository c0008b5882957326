"""Controller GRU: stock nn.GRU (MIOpen) vs the persistent HIP recurrence, forward and forward+backward."""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")  # kernel-form / tiling hooks (include/ddsp_hip.h)
import sys, time, torch, torch.nn as nn
sys.path.insert(0, '.')
import ddsp_pytorch_amd as ddsp

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

for (B, T, n_in, hd) in [(32, 500, 1024, 512), (1, 32, 1024, 512), (1, 16, 1024, 512), (128, 500, 1024, 512)]:
    torch.manual_seed(0)
    ref = nn.GRU(n_in, hd, 1, batch_first=True).cuda()
    mine = ddsp.GRU(n_in, hd, 1, batch_first=True).cuda(); mine.load_state_dict(ref.state_dict())
    x = torch.randn(B, T, n_in, device='cuda'); xg = x.clone().requires_grad_(True)
    h0 = torch.randn(1, B, hd, device='cuda')
    def fwd(m):
        with torch.no_grad(): return m(x, h0)
    def fb(m):
        y, h = m(xg, h0); (y.square().mean() + h.mean()).backward()
    out = {"shape": (B, T, n_in, hd)}
    for name, m in (("miopen", ref), ("hip", mine)):
        out[name + "_fwd_ms"] = round(timeit(lambda: fwd(m)), 3)
        out[name + "_fwd_bwd_ms"] = round(timeit(lambda: fb(m)), 3)
    with torch.no_grad():
        d = float((ref(x, h0)[0] - mine(x, h0)[0]).abs().max())
    out["max_abs_diff_vs_miopen"] = d
    print(out, flush=True)

# recurrence only (no input-projection GEMM): microseconds per time step
from ddsp_pytorch_amd import gru as G
import os
from ddsp_pytorch_amd import _lib
_lib.lib().ddsp_gru_set_mode(int(os.environ.get('GRU_MODE', '0')))
for (B, T, hd) in [(32, 500, 512), (1, 32, 512), (8, 500, 512), (64, 500, 512), (32, 500, 128)]:
    gi = torch.randn(B, T, 3 * hd, device='cuda'); w = torch.randn(3 * hd, hd, device='cuda') * 0.05
    b = torch.zeros(3 * hd, device='cuda'); h0 = torch.zeros(B, hd, device='cuda')
    f = timeit(lambda: G.gru_forward(gi, w, b, h0, save=True))
    y, hT, gates, hn = G.gru_forward(gi, w, b, h0, save=True)
    dy = torch.randn_like(y)
    bw = timeit(lambda: G.gru_backward(dy, None, w, h0, y, gates, hn))
    print({"mode": os.environ.get("GRU_MODE", "0"), "recurrence": (B, T, hd), "fwd_us_per_step": round(f * 1e3 / T, 2), "bwd_us_per_step": round(bw * 1e3 / T, 2)}, flush=True)
