import torch, json
x = torch.randn(32, 64000, device="cuda")
out = {}
for n in (128000, 131072, 81920, 98304, 80000, 82944, 86400, 96000, 100000, 102400, 65536 * 2, 90112, 87808, 84000, 83200, 80640):
    for _ in range(3):
        s = torch.fft.rfft(x, n=n); y = torch.fft.irfft(s, n=n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        s = torch.fft.rfft(x, n=n)
    e1.record(); torch.cuda.synchronize(); tf = e0.elapsed_time(e1) / 20
    e0.record()
    for _ in range(20):
        y = torch.fft.irfft(s, n=n)
    e1.record(); torch.cuda.synchronize(); ti = e0.elapsed_time(e1) / 20
    out[n] = (round(tf * 1e3, 1), round(ti * 1e3, 1))
print(json.dumps(out))
