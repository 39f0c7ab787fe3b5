#!/usr/bin/env python3
"""How long does hipMalloc / hipFree take for tens of GB on this box?  (The 3.1 Gbp index build spends most of its wall
time there: tests/tools/scale_check.py with SLAMEM_BUILD_TRACE=1.)"""
import ctypes as C
import time

import torch  # loads the HIP runtime the same way the engine does

hip = C.CDLL("libamdhip64.so")
torch.cuda.init()


def malloc(gb):
    p = C.c_void_p()
    t = time.time()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(int(gb * (1 << 30))))
    dt = time.time() - t
    assert rc == 0, rc
    return p, dt


def free(p):
    t = time.time()
    hip.hipFree(p)
    return time.time() - t


for gb in (1, 10, 40, 80):
    p, dt = malloc(gb)
    df = free(p)
    p2, dt2 = malloc(gb)
    hip.hipMemset(p2, 0, C.c_size_t(int(gb * (1 << 30))))
    hip.hipDeviceSynchronize()
    df2 = free(p2)
    print(f"{gb:3d} GiB: hipMalloc {dt * 1e3:8.1f} ms, hipFree {df * 1e3:7.1f} ms, again hipMalloc {dt2 * 1e3:8.1f} ms, hipFree after use {df2 * 1e3:7.1f} ms")
ps = []
t = time.time()
for _ in range(8):
    ps.append(malloc(10)[0])
print(f"8 x 10 GiB: {1e3 * (time.time() - t):.1f} ms")
for p in ps:
    free(p)
