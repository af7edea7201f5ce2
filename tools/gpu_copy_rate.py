#!/usr/bin/env python3
"""Calibration: device-to-device copy and read-only reduction rates of this box (what 'HBM-bound' can mean here)."""
import torch
dev = torch.device('cuda', 0)
for mb in (256, 1024, 2048):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.float16, device=dev).normal_()
    y = torch.empty_like(x)
    for name, fn, bytes_ in (('copy', lambda: y.copy_(x), 2 * n * 2), ('read(sum)', lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32), n * 2),
                             ('write(fill)', lambda: y.fill_(1.0), n * 2), ('add3', lambda: torch.add(x, y, out=y), 3 * n * 2)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print('%5d MB %-12s %.3f ms  %.2f TB/s' % (mb, name, best, bytes_ / best / 1e9))
