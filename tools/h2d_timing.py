import time, torch, numpy as np
n = 26_500_000 * 10
x = torch.randint(-3000, 3000, (n,), dtype=torch.int16)
y = torch.empty(n, dtype=torch.int16, device="cuda")
xp = x.pin_memory()
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
for m in (n // 10, n):
    a = t(lambda: y[:m].copy_(x[:m]))
    b = t(lambda: y[:m].copy_(xp[:m], non_blocking=True))
    z = torch.empty(m, dtype=torch.int16)
    c = t(lambda: z.copy_(x[:m]))
    zp = torch.empty(m, dtype=torch.int16).pin_memory()
    d = t(lambda: zp.copy_(x[:m]))
    print("bytes %d MB: pageable H2D %.1f GB/s, pinned H2D %.1f GB/s, host copy (torch, threads=%d) %.1f GB/s, host copy into pinned %.1f GB/s" % (m * 2 >> 20, m * 2 / a / 1e9, m * 2 / b / 1e9, torch.get_num_threads(), m * 2 / c / 1e9, m * 2 / d / 1e9))
torch.set_num_threads(1)
m = n // 10
z = torch.empty(m, dtype=torch.int16)
print("one thread host copy %.1f GB/s" % (m * 2 / t(lambda: z.copy_(x[:m])) / 1e9))
