#!/usr/bin/env python3
"""Probe (GPU box): what a dependency between two HIP streams of one device costs, per mechanism. Main stream: 40 kernels of ~50 us back to
back; behind every second one the side stream is released to run a ~40 us kernel. Variants: no dependency at all (floor), torch events,
raw HIP events without timing / system fence (unet._Side today), hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory."""
import ctypes, time, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd import _lib
dev = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
b = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
c = torch.randn(1536, 2048, device=dev, dtype=torch.bfloat16)
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
h = _lib._hip_runtime()
h.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
h.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
h.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
sig = ctypes.c_void_p()
rc = h.hipExtMallocWithFlags(ctypes.byref(sig), 8, 0x2)
print("hipExtMallocWithFlags(signal, 8 bytes) rc", rc)
if rc != 0:
    h.hipGetLastError()
    flag_t = torch.zeros(2, dtype=torch.int32, device=dev)
    sig = ctypes.c_void_p(flag_t.data_ptr())
    rc = 0
    print("falling back to plain device memory for the flag")
epoch = [0]

def run(mode, n=40):
    for i in range(n):
        torch.mm(a, b)
        if i % 2 == 1 and mode != "none":
            if mode == "torch":
                ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            elif mode == "raw":
                _lib.hip_fork(main.cuda_stream, side.cuda_stream, 0)
            elif mode == "value":
                epoch[0] += 1
                r1 = h.hipStreamWriteValue32(ctypes.c_void_p(main.cuda_stream), sig, epoch[0], 0)
                r2 = h.hipStreamWaitValue32(ctypes.c_void_p(side.cuda_stream), sig, epoch[0], 0, 0xffffffff)
                assert r1 == 0 and r2 == 0, (r1, r2)
        if i % 2 == 1:
            with torch.cuda.stream(side):
                torch.mm(c, b)
    main.wait_stream(side)

for mode in ("none", "torch", "raw", "value", "none", "torch", "raw", "value"):
    if mode == "value" and rc != 0:
        continue
    run(mode, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run(mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-6s %.1f us per pass of 40 main + 20 side kernels" % (mode, dt * 1e6))
