import time, torch
torch.set_num_threads(16)
x = torch.rand(512, 500, 128); y = x.cuda(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): x = torch.rand(512, 500, 128)
t1 = time.perf_counter()
for _ in range(5): y = x.cuda()
torch.cuda.synchronize(); t2 = time.perf_counter()
xp = x.pin_memory()
t3 = time.perf_counter()
for _ in range(5): y = xp.cuda(non_blocking=True)
torch.cuda.synchronize(); t4 = time.perf_counter()
print("torch.rand(512,500,128) on CPU: %.1f ms ; pageable H2D: %.1f ms (%.1f GB/s) ; pinned H2D: %.1f ms (%.1f GB/s)" % (
    (t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3, 0.131072 / ((t2 - t1) / 5), (t4 - t3) / 5 * 1e3, 0.131072 / ((t4 - t3) / 5)))
