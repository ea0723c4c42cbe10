"""HBM write / copy bandwidth of plain torch kernels on this device (what a write-bound kernel can hope for)."""
import time
import torch
x = torch.empty(1 << 28, device="cuda", dtype=torch.float32)      # 1 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n
gb = x.numel() * 4 / 1e9
print("fill_  (write only): %.2f TB/s" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("copy_  (read+write): %.2f TB/s total" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
print("sum    (read only) : %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
h = torch.empty(1 << 28, device="cuda", dtype=torch.bfloat16)
print("bf16 fill_ 0.5 GiB : %.2f TB/s" % (h.numel() * 2 / 1e9 / t(lambda: h.fill_(1.0)) / 1e3))
