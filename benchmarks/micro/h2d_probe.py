#!/usr/bin/env python3
"""How to get a (262144, 3) float32 host array onto the GPU fastest: through a
pinned staging buffer (np.copyto + async copy) or straight from pageable memory
(the runtime's own chunked staging)?  Times until the data is usable on the device."""
import time

import numpy as np
import torch

n = 262144
x = np.random.rand(n, 3).astype(np.float32)
dev = torch.empty((n, 3), dtype=torch.float32, device='cuda:0')
pin = torch.empty((n, 3), dtype=torch.float32).pin_memory()
pin_np = pin.numpy()


def pinned():
    np.copyto(pin_np, x)
    dev.copy_(pin, non_blocking=True)


def pageable():
    dev.copy_(torch.from_numpy(x), non_blocking=False)


def pinned_only():
    dev.copy_(pin, non_blocking=True)


def host_copy_only():
    np.copyto(pin_np, x)


for name, fn in (('pinned staging (host copy + async H2D)', pinned),
                 ('pageable direct', pageable),
                 ('async H2D from pinned alone', pinned_only),
                 ('host copy into pinned alone', host_copy_only)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
        torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms')
d2h = torch.empty((n, 3), dtype=torch.float32).pin_memory()
for name, fn in (('D2H to pageable (.to(cpu))', lambda: dev.to('cpu')),
                 ('D2H to pinned', lambda: d2h.copy_(dev, non_blocking=True))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
        torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms')
