#!/usr/bin/env python3
"""Calibrate achievable HBM bandwidth on the box with plain torch kernels (copy: 1R+1W, add: 2R+1W)."""
import torch
dev = torch.device('cuda:0')
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.randn(n, device=dev); z = torch.empty_like(x)
    for name, fn, nbytes in (('copy', lambda: z.copy_(x), 2 * n * 4), ('add ', lambda: torch.add(x, y, out=z), 3 * n * 4),
                             ('read', lambda: x.sum(), n * 4), ('fill', lambda: z.fill_(1.0), n * 4)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('%5d MB %s: %.3f ms  %.2f TB/s' % (mb, name, ms, nbytes / ms / 1e9))
