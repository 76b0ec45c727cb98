#!/usr/bin/env python3
"""Model of the conv tile planner (rsu_api.hip plan_fwd2 / plan_geo_aligned): pixel / total utilisation of the 3x3
forward and backward-data launches of a configuration, for strip-width policies. CPU only (no GPU needed)."""
import sys, math
def cdiv(a, b): return (a + b - 1) // b
def rup(a, b): return cdiv(a, b) * b

CFGS = {  # name: (TN, TM, pt per wave)
    "128x256": (128, 256, 4), "64x512": (64, 512, 4), "128x128": (128, 128, 2), "64x256": (64, 256, 2),
    "128x192": (128, 192, 3), "64x384": (64, 384, 3), "128x320": (128, 320, 5), "64x640": (64, 640, 5)}

def layers_c2(L=5, root=64, P=388):
    # (name, Hout, Cin, Cout) of every 3x3 conv, forward geometry
    o = P
    for _ in range(L - 1): o = (o + 4) // 2
    for _ in range(L - 1): o = (o + 4) * 2
    S = o + 4
    out = []
    h, c = S, 3
    for l in range(L):
        co = root << l
        out.append((f"conv_{l}/conv1", h - 2, c if l else 16, co)); h -= 2
        out.append((f"conv_{l}/conv2", h - 2, co, co)); h -= 2
        c = co
        if l < L - 1: h //= 2
    for l in range(L - 1):
        co = c // 2
        h *= 2
        out.append((f"conv_{L + l}/conv1", h - 2, c, co)); h -= 2
        out.append((f"conv_{L + l}/conv2", h - 2, co, co)); h -= 2
        c = co
    return out

def geo_options(Ho, Wo, TM, policy):
    opts = []
    if policy == "pow2":
        sws = [8, 16, 32, 64]
    else:
        sws = list(range(8, 257, 8))
    for SW in sws:
        if SW > TM: break
        TR = TM // SW
        if TR < 1: continue
        if policy == "pow2" and TR * SW != TM: continue
        CW = rup(SW + 2, 8)
        npix = rup((TR + 2) * CW, 32)
        if npix * 64 * 2 > 100 * 1024: continue   # halo ring budget (rough)
        tiles = cdiv(Wo, SW) * cdiv(Ho, TR)
        opts.append((tiles, npix / TM, SW, TR))
    return opts

def plan(N, Ho, Wo, Cout, K, ncu, policy, shapes):
    best = None
    for name in shapes:
        TN, TM, pt = CFGS[name]
        if Cout <= 64 and TN > 64: continue
        if Cout > 64 and TN <= 64: continue
        for tiles, halo, SW, TR in geo_options(Ho, Wo, TM, policy):
            ncob = cdiv(Cout, TN)
            ntile = N * tiles
            workers = max(1, min(ncu // ncob, ntile))
            rounds = cdiv(ntile, workers)
            ovh = min(max(4096.0 * 1152 / K, 4096.0), 65536.0)
            eff = {2: 1.5, 3: 1.2, 5: 0.96}.get(pt, 1.0)
            cost = rounds * (TM * TN * eff + ovh) * (1 + 0.02 * halo)
            useful = N * Ho * Wo * Cout
            cand = (cost, name, SW, TR, ntile, workers * ncob, rounds, useful / (ntile * TM * ncob * TN), useful / (rounds * workers * ncob * TM * TN))
            if best is None or cand < best: best = cand
    return best

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    shapes_pp = ["128x256", "64x512", "128x128", "64x256", "128x192", "64x384", "128x320", "64x640"]
    for ncu in (256, 128):
        for policy in ("pow2", "mult8"):
            tot_u = tot_w = 0
            print(f"--- N={N} ncu={ncu} policy={policy}")
            for (name, H, ci, co) in layers_c2():
                if ci == 16: continue
                for op, (Ho, Cn, K) in (("fwd", (H, co, 9 * ci)), ("bwd", (H + 2, ci, 9 * co))):
                    b = plan(N, Ho, Ho, Cn, K, ncu, policy, shapes_pp)
                    gf = 2.0 * N * Ho * Ho * Cn * K / 1e9
                    print(f"{name:14s} {op} {Ho:4d} C{Cn:5d} K{K:6d} {gf:6.1f}GF  {b[1]:8s} SW{b[2]:3d} TR{b[3]:3d} tiles{b[4]:5d} grid{b[5]:4d} rounds{b[6]:3d} pix_util {b[7]:.3f} total_util {b[8]:.3f}")
                    tot_u += gf; tot_w += gf / b[8]
            print(f"weighted total_util {tot_u / tot_w:.3f}")
