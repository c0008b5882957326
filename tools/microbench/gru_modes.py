"""Recurrence only: groups kept on one XCD (mode 0, default) vs dealt over all XCDs (mode 1)."""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")  # kernel-form / tiling hooks (include/ddsp_hip.h)
import sys, time, torch
sys.path.insert(0, '.')
from ddsp_pytorch_amd import gru as G, _lib
L = _lib.lib()
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for mode in (0, 1, 0, 1):
    L.ddsp_gru_set_mode(mode)
    for (B, T, hd) in [(16, 500, 512), (4, 500, 512), (1, 500, 512)]:
        gi = torch.randn(B, T, 3 * hd, device='cuda'); w = torch.randn(3 * hd, hd, device='cuda') * 0.05
        b = torch.zeros(3 * hd, device='cuda'); h0 = torch.zeros(B, hd, device='cuda')
        f = timeit(lambda: G.gru_forward(gi, w, b, h0, save=True))
        y, hT, gates, hn = G.gru_forward(gi, w, b, h0, save=True)
        dy = torch.randn_like(y)
        bw = timeit(lambda: G.gru_backward(dy, None, w, h0, y, gates, hn))
        print({"mode": mode, "recurrence": (B, T, hd), "fwd_us_per_step": round(f * 1e3 / T, 2), "bwd_us_per_step": round(bw * 1e3 / T, 2)}, flush=True)
L.ddsp_gru_set_mode(0)
