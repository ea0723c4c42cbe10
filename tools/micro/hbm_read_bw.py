"""What a plain streaming READ reaches on this device (the yardstick for the weight-gradient products, which only read):
torch.sum over 4 GiB of fp32 / bf16, and a device copy (read + write)."""
import time

import torch

x = torch.ones(2 ** 30, dtype=torch.float32, device="cuda")
for name, fn, nbytes in (("sum fp32 4 GiB", lambda: x.sum(), x.numel() * 4),
                         ("sum bf16 view", lambda: x.view(torch.bfloat16).sum(), x.numel() * 4),
                         ("copy 4 GiB (read + write)", lambda: torch.empty_like(x).copy_(x), x.numel() * 8)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("%-28s %.3f ms  %.2f TB/s" % (name, dt * 1e3, nbytes / dt / 1e12))
