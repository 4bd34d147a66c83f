import time, numpy as np, torch
src = np.random.randint(0, 255, (480, 640), np.uint8)
t = torch.empty((480, 640), dtype=torch.uint8, device="cuda")
pin = torch.empty((480, 640), dtype=torch.uint8).pin_memory()
ts = torch.from_numpy(src)
def med(fn, n=200):
    fn(); torch.cuda.synchronize()
    xs = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); xs.append(time.perf_counter() - t0)
    return 1e6 * sorted(xs)[n // 2]
print("pageable H2D 307 KB: %.1f us" % med(lambda: t.copy_(ts)))
print("memcpy into pinned + H2D: %.1f us" % med(lambda: (pin.copy_(ts), t.copy_(pin, non_blocking=True))))
print("pinned H2D alone: %.1f us" % med(lambda: t.copy_(pin, non_blocking=True)))
out = torch.empty((1100, 60), dtype=torch.uint8, device="cuda"); hout = torch.empty((1100, 60), dtype=torch.uint8).pin_memory()
print("D2H 66 KB pinned: %.1f us" % med(lambda: hout.copy_(out, non_blocking=True)))
