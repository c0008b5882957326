"""A few forward + backward launches of the GRU recurrence at the training shape, for rocprofv3 (kernel trace / PMC)."""
import sys, torch
sys.path.insert(0, '.')
from ddsp_pytorch_amd import gru as G
B, T, hd = 32, 500, 512
torch.manual_seed(0)
gi = torch.randn(B, T, 3 * hd, device='cuda'); w = torch.randn(3 * hd, hd, device='cuda') * 0.05
b = torch.zeros(3 * hd, device='cuda'); h0 = torch.zeros(B, hd, device='cuda')
for _ in range(6):
    y, hT, gates, hn = G.gru_forward(gi, w, b, h0, save=True)
    dy = torch.randn_like(y)
    G.gru_backward(dy, None, w, h0, y, gates, hn)
torch.cuda.synchronize()
print("ok")
