"""Host->device copy rates on this box: one large pinned copy vs many 1 MiB pinned copies vs pageable."""
import time, torch
n = 256 * 131072 * 2  # floats = 268 MB
dev = torch.empty(n, dtype=torch.float32, device="cuda")
pin = torch.empty(n, dtype=torch.float32).pin_memory()
pag = torch.empty(n, dtype=torch.float32)
pag.fill_(1.0); pin.fill_(2.0)
def t(f, k=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k
gb = n * 4 / 1e9
print("one pinned copy     : %.1f GB/s" % (gb / t(lambda: dev.copy_(pin, non_blocking=True))))
dv = dev.view(256, -1); pv = pin.view(256, -1)
print("256 pinned copies   : %.1f GB/s" % (gb / t(lambda: [dv[i].copy_(pv[i], non_blocking=True) for i in range(256)])))
print("one pageable copy   : %.1f GB/s" % (gb / t(lambda: dev.copy_(pag))))
s2 = torch.cuda.Stream()
half = n // 2
def two():
    dev[:half].copy_(pin[:half], non_blocking=True)
    with torch.cuda.stream(s2):
        dev[half:].copy_(pin[half:], non_blocking=True)
    s2.synchronize()
print("two streams, halves : %.1f GB/s" % (gb / t(two)))
out = torch.empty(n // 8, dtype=torch.float32).pin_memory()
print("D2H pinned 33 MB    : %.1f GB/s" % (gb / 8 / t(lambda: out.copy_(dev[:n // 8], non_blocking=True))))
