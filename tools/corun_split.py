#!/usr/bin/env python3
"""Developer tool (GPU box): split the co-run penalty of the two backward queues into power and traffic.
The 17 backward-data launches of c2 (planned for 128 CUs, main stream) -- and, the other way round, the 17 weight-gradient launches -- are
timed alone, beside a register-only MFMA burner on the other 128 CUs (power, no traffic), beside a pure streamer (traffic, no MFMA) and beside
the real other family; every situation runs as a sustained loop (~2 s) while the socket power is sampled, and the synthetic neighbours note
the shader clock they ran at (s_memtime over s_memrealtime). Needs probes/libprobe_corun.so (probes/probe_corun.hip).
usage: python tools/corun_split.py [--seconds 2.0] [--B 4]  -> stdout (tee it to gpurun_out/r06/corun_split.txt)"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from road_segmentation_unet_amd._lib import RsuSrc, call, lib  # noqa: E402
from tools.bench_layers import layers  # noqa: E402

DEV = "cuda:0"
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Power(threading.Thread):
    """socket power sampled from the hwmon file of the device (falls back to rocm-smi), every ~50 ms"""

    def __init__(self):
        super().__init__(daemon=True)
        pr = torch.cuda.get_device_properties(0)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        cands = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*/power1_average" % bdf) + glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*/power1_input" % bdf)
        self.path = None
        for c in cands:
            try:
                if int(open(c).read()) > 0:
                    self.path = c
                    break
            except (OSError, ValueError):
                pass
        self.samples, self.on, self.stop_ = [], False, False

    def read(self):
        if self.path:
            try:
                return int(open(self.path).read()) / 1e6
            except (OSError, ValueError):
                return None
        try:
            out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            for ln in out.splitlines():
                if "Power" in ln and ":" in ln:
                    return float(ln.split(":")[-1].strip())
        except Exception:
            return None
        return None

    def run(self):
        while not self.stop_:
            if self.on:
                v = self.read()
                if v:
                    self.samples.append(v)
            time.sleep(0.05 if self.path else 0.01)

    def begin(self):
        self.samples, self.on = [], True

    def end(self):
        self.on = False
        s = self.samples[len(self.samples) // 4:]   # (the first quarter: the averaging window still holds the situation before)
        return (float(np.median(s)), len(s)) if s else (float("nan"), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--ncu", type=int, default=128)
    ap.add_argument("--quick", action="store_true", help="only: alone, the weakest burner and streamer, the real neighbour")
    args = ap.parse_args()
    P = ctypes.CDLL(os.path.join(ROOT, "probes", "libprobe_corun.so"))
    P.corun_burn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    P.corun_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    P.corun_clock.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p]
    main_s = torch.cuda.current_stream()
    side_s = torch.cuda.Stream()
    clk_s = torch.cuda.Stream()
    st = ctypes.c_void_p(main_s.cuda_stream)
    sst = ctypes.c_void_p(side_s.cuda_stream)
    call("rsu_set_autotune", 2)
    nk = int(lib().rsu_conv_splitk_ws_floats())
    kws = torch.zeros(nk, device=DEV)
    ncu, B = args.ncu, args.B
    bwd, wg, flops = [], [], 0.0
    keep = []
    for name, hin, cin, cout, dil in layers(5, 64, 388):
        ho = hin - 2 * dil
        x = torch.randn((B, hin, hin, cin), device=DEV).to(torch.bfloat16)
        dz = torch.randn((B, ho, ho, cout), device=DEV).to(torch.bfloat16)
        dx = torch.zeros_like(x)
        w = torch.randn((3, 3, cin, cout), device=DEV) * 0.05
        seg2 = (ctypes.c_int * 1)(cout)
        pb = torch.zeros(lib().rsu_packed_bytes(9, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=DEV)
        call("rsu_pack_conv_bwd", ptr(w), ptr(pb), 3, cin, 0, cin, cout, st)
        dw = torch.zeros_like(w)
        db = torch.zeros(cout, device=DEV)
        ws = torch.zeros(lib().rsu_conv2d_bwd_weight_ws_floats(cin, cin, cout), device=DEV)
        src = RsuSrc(x.data_ptr(), hin, hin, cin, 0, 0)
        keep.append((x, dz, dx, w, pb, dw, db, ws, src))
        flops += 2.0 * B * ho * ho * cout * cin * 9
        bwd.append(lambda s, dz=dz, pb=pb, dx=dx, x=x, hin=hin, cin=cin, cout=cout, dil=dil: call(
            "rsu_conv2d_bwd_data_k", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, hin, hin, cin, 0, cin, cout, dil, ncu, ptr(kws), nk, s))
        wg.append(lambda s, src=src, dz=dz, dw=dw, db=db, ws=ws, ho=ho, cin=cin, cout=cout, dil=dil: call(
            "rsu_conv2d_bwd_weight", ctypes.byref(src), ptr(dz), ptr(dw), ptr(db), ptr(ws), B, ho, ho, cin, 0, cout, dil, ncu, s))
    seqs = {"backward-data": bwd, "weight-gradient": wg}
    for fs in seqs.values():   # tuning pass + warm-up on an idle device
        for _ in range(2):
            for f in fs:
                f(st)
    torch.cuda.synchronize()

    rnd = (torch.randn(4096 * 8, device=DEV)).to(torch.bfloat16)
    outb = torch.zeros(1024, device=DEV)
    clk = torch.zeros(512, dtype=torch.int64, device=DEV)
    clk2 = torch.zeros(64, dtype=torch.int64, device=DEV)      # the idle clock samplers' (free CUs: only with --ncu <= 124)
    have_free = 2 * ncu <= 248
    NSL, N16 = 4096, 16384                       # 4096 slices of 256 KiB = 1 GiB: nothing is re-read from L2 / the Infinity Cache
    big = torch.empty(NSL * N16 * 16, dtype=torch.uint8, device=DEV).random_(0, 255)
    big2 = torch.empty_like(big)
    nwg = min(256 - ncu, ncu)

    def burn(rounds, sleep, nread=0):
        return lambda s: P.corun_burn(ptr(rnd), ptr(outb), ptr(clk), nwg, rounds, sleep, nread, s)

    def stream(passes, mode, sleep):
        return lambda s: P.corun_stream(ptr(big), ptr(big2), ptr(clk), nwg, N16, passes, NSL, mode, sleep, s)

    def time_alone(fn, s_t, s_c, reps=3):
        fn(s_c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_t):
            e0.record()
            for _ in range(reps):
                fn(s_c)
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    def clock_of():
        c = clk.cpu().numpy().reshape(-1, 2)[:nwg].astype(np.float64)
        ok = c[:, 1] > 0
        return float(np.median(c[ok, 0] / c[ok, 1]) * 0.1) if ok.any() else float("nan")

    power = Power()
    power.start()
    print("power source: %s" % (power.path or "rocm-smi"), flush=True)

    def free_clock():
        c = clk2.cpu().numpy().reshape(-1, 2)[:8].astype(np.float64)
        ok = c[:, 1] > 0
        return float(np.median(c[ok, 0] / c[ok, 1]) * 0.1) if (have_free and ok.any()) else float("nan")

    def sustained(main_fns, side_fn, seconds, t_guess, t_clock=0.0):
        """main_fns on the main stream (timed, events per repetition) beside side_fn on the side stream; returns (median ms, power W, n)"""
        reps = max(8, int(seconds / t_guess))
        evs = []
        power.begin()
        for _ in range(reps):
            if have_free and t_clock:
                clk_s.wait_stream(main_s)
                P.corun_clock(ptr(clk2), 8, int(t_clock * 1e8), ctypes.c_void_p(clk_s.cuda_stream))
            if side_fn is not None:
                side_s.wait_stream(main_s)
                side_fn(sst)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            for f in main_fns:
                f(st)
            e1.record(main_s)
            if side_fn is not None:
                main_s.wait_stream(side_s)
            if have_free and t_clock:
                main_s.wait_stream(clk_s)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        w, nsmp = power.end()
        ts = [a.elapsed_time(b) for a, b in evs[2:]]
        return float(np.median(ts)), w, nsmp

    # ---- the synthetic neighbours alone: rate, clock, power
    print("== neighbours alone on %d workgroups (one per CU)" % nwg, flush=True)
    burn_cfg = {}
    BURNS = ((0, 0), (2, 0), (4, 0), (8, 0), (0, 3), (2, 3), (4, 3))     # (sleep, nread): nread 3 = 24 ds_read_b128 per 48 MFMAs, a conv's rate
    if args.quick:
        BURNS = ((8, 0),)
    for sleep, nread in BURNS:
        R = 2000
        t = time_alone(burn(R, sleep, nread), main_s, st)
        tf = nwg * 8 * R * 48 * 2.0 * 16 * 16 * 32 / t / 1e12
        ms, w, n = sustained([burn(R, sleep, nread)], None, 1.0, t)
        burn_cfg[(sleep, nread)] = (R / t, tf)
        print("burner sleep %d, %2d LDS reads / 48 MFMAs: %4.0f TFLOP/s on %d CUs (%.1f %% of their share of 2.5 PF), clock %.3f GHz, power %.0f W (%d samples)" % (
            sleep, 8 * nread, tf, nwg, 100 * tf / (2500.0 * nwg / 256), clock_of(), w, n), flush=True)
    stream_cfg = {}
    STREAMS = ((0, 0), (0, 4), (0, 16), (0, 32), (0, 64), (1, 0), (1, 32), (1, 64))
    if args.quick:
        STREAMS = ((0, 64),)
    for mode, sleep in STREAMS:
        passes = 8
        t = time_alone(stream(passes, mode, sleep), main_s, st)
        gbs = nwg * passes * N16 * 16 * (2 if mode else 1) / t / 1e9
        ms, w, n = sustained([stream(passes, mode, sleep)], None, 0.6, t)
        stream_cfg[(mode, sleep)] = (passes / t, gbs)
        print("streamer mode %s sleep %2d: %5.0f GB/s, clock %.3f GHz, power %.0f W" % ("copy" if mode else "read", sleep, gbs, clock_of(), w), flush=True)

    # an idle-clock sample: eight idle workgroups while nothing else runs
    P.corun_clock(ptr(clk), 8, 100000, st)
    torch.cuda.synchronize()
    c = clk.cpu().numpy().reshape(-1, 2)[:8].astype(np.float64)
    print("idle workgroups (nothing else running): clock %.3f GHz" % float(np.median(c[:, 0] / c[:, 1]) * 0.1), flush=True)

    for mname, oname in (("backward-data", "weight-gradient"), ("weight-gradient", "backward-data")):
        fns, other = seqs[mname], seqs[oname]
        t_alone = time_alone(lambda s: [f(s) for f in fns], main_s, st)
        print("== %s launches (17, planned for %d CUs) on the main stream; %.1f GFLOP; neighbours on %d CUs%s" % (
            mname, ncu, flops / 1e9, nwg, "; 8 idle clock samplers on free CUs" if have_free else ""), flush=True)
        rows = []
        tc = 0.9 * t_alone

        def add(label, side, guess, nclk=True):
            ms, w, n = sustained(fns, side, args.seconds, guess, tc)
            rows.append((label, ms, w, clock_of() if (side is not None and nclk) else float("nan"), free_clock()))
            base = rows[0][1]
            print("  %-66s %7.3f ms  %+6.1f %%  %5.0f TFLOP/s (x 256/%d)  power %5.0f W  neighbour clock %s  free-CU clock %s" % (
                label, ms, 100 * (ms / base - 1), (256.0 / ncu) * flops / (ms * 1e-3) / 1e12, ncu, w,
                ("%.3f GHz" % rows[-1][3]) if rows[-1][3] == rows[-1][3] else "--", ("%.3f GHz" % rows[-1][4]) if rows[-1][4] == rows[-1][4] else "--"), flush=True)
        add("alone (other CUs idle)", None, t_alone)
        target = t_alone * 1.45   # the neighbour must outlast the measured sequence, co-run penalty included
        for sleep, nread in BURNS:
            rps, tf = burn_cfg[(sleep, nread)]
            add("beside MFMA burner sleep %d, %2d LDS reads (%.0f TF alone)" % (sleep, 8 * nread, tf), burn(int(rps * target), sleep, nread), target)
        for mode, sleep in STREAMS:
            pps, gbs = stream_cfg[(mode, sleep)]
            add("beside streamer %s sleep %2d (%.0f GB/s alone)" % ("copy" if mode else "read", sleep, gbs), stream(max(1, int(pps * target)), mode, sleep), target)

        def real(s):
            for _ in range(2):
                for f in other:
                    f(s)
        add("beside the real %s launches (two passes of 17)" % oname, real, 2.6 * t_alone, nclk=False)
    power.stop_ = True


if __name__ == "__main__":
    main()
