#!/usr/bin/env python3
"""Developer tool (GPU box): which XCD / SE / CU do the workgroups of a launch land on -- on the default stream and on
streams made with hipExtStreamCreateWithCUMask for several mask patterns? Answers whether a stream can be given XCDs of its own.
Needs probes/libprobe_corun.so. usage: python tools/cumask_where.py"""
import collections
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from road_segmentation_unet_amd import _lib  # noqa: E402

P = ctypes.CDLL(os.path.join(ROOT, "probes", "libprobe_corun.so"))
P.corun_whereami.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p]
h = _lib._hip_runtime()
h.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
torch.zeros(1, device="cuda:0")
out = torch.zeros(4096, dtype=torch.int32, device="cuda:0")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(bits[w * 32 + b] << b for b in range(32)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = h.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    return rc, s


def where(stream, nwg, lds):
    out.zero_()
    rc = P.corun_whereami(ctypes.c_void_p(out.data_ptr()), nwg, 3000, lds, stream)   # 30 us per workgroup
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.int64).reshape(-1, 2)[:nwg] & 0xFFFFFFFF
    xcc = o[:, 0] & 0xF
    cu, sh, se = (o[:, 1] >> 8) & 0xF, (o[:, 1] >> 12) & 1, (o[:, 1] >> 13) & 7
    per = collections.Counter(xcc.tolist())
    cus = {x: len(set(zip(se[xcc == x].tolist(), sh[xcc == x].tolist(), cu[xcc == x].tolist()))) for x in sorted(per)}
    rr = "".join(str(int(v)) for v in xcc[:24])
    return rc, dict(sorted(per.items())), cus, rr


patterns = collections.OrderedDict()
patterns["all 256 bits"] = [1] * 256
patterns["first 128 bits"] = [1 if i < 128 else 0 for i in range(256)]
patterns["first 64 bits"] = [1 if i < 64 else 0 for i in range(256)]
patterns["bits with i % 8 < 4 (XCDs 0-3 if bit i belongs to XCD i % 8)"] = [1 if i % 8 < 4 else 0 for i in range(256)]
patterns["bits with i % 8 >= 4"] = [1 if i % 8 >= 4 else 0 for i in range(256)]
patterns["bits with i % 8 == 0"] = [1 if i % 8 == 0 else 0 for i in range(256)]
patterns["bits 0-31 (XCD 0 if bits are grouped by XCD)"] = [1 if i < 32 else 0 for i in range(256)]
for nwg, lds in ((256, 150 * 1024), (128, 150 * 1024), (256, 0)):
    print("== %d workgroups of 64 threads, %d KB of LDS each (resident together for 30 us)" % (nwg, lds // 1024))
    rc, per, cus, rr = where(None, nwg, lds)
    print("  %-70s rc %d | workgroups per XCD %s | distinct CUs used per XCD %s | XCD of workgroups 0..23: %s" % ("default stream", rc, per, cus, rr))
    for name, bits in patterns.items():
        rc, s = masked_stream(bits)
        if rc != 0:
            print("  %-70s hipExtStreamCreateWithCUMask rc %d" % (name, rc))
            continue
        rc2, per, cus, rr = where(s, nwg, lds)
        print("  %-70s rc %d | workgroups per XCD %s | distinct CUs used per XCD %s | XCD of workgroups 0..23: %s" % (name, rc2, per, cus, rr), flush=True)
