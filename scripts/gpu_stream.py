"""Measured HBM ceilings on this box: device-to-device copy and y += a*x (the library's own axpy kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from athena_amd import ops, _capi
_capi.init(0)
n = 1 << 28   # 1 GiB per vector
x = torch.rand(n, device="cuda"); y = torch.rand(n, device="cuda")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ms = t(lambda: y.copy_(x)); print("copy   : %.3f ms  %.2f TB/s (read+write)" % (ms, 2 * 4 * n / ms / 1e9))
ms = t(lambda: ops.axpy(0.5, x, y)); print("axpy   : %.3f ms  %.2f TB/s (2 reads + 1 write)" % (ms, 3 * 4 * n / ms / 1e9))
ms = t(lambda: torch.sum(x)); print("read   : %.3f ms  %.2f TB/s (read only)" % (ms, 4 * n / ms / 1e9))
