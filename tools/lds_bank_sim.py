"""Bank-conflict simulator for the LDS images used by the igemm kernels (developer tool).
Rules from /opt/skills/guides/MI355X_MICROARCH.md section LDS: ds_read_b128 is served in four 16-lane
groups, bank = (addr/4) % 64; ds_read_b64_tr_b16 in two 32-lane halves, bank = (addr/4) % 64."""
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALVES = [list(range(0, 32)), list(range(32, 64))]


def cycles(addrs, groups, nbytes):
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs[l]
            for b in range(nbytes // 4):
                bank = ((a // 4) + b) % 64
                banks.setdefault(bank, set()).add(a + 4 * b)
        tot += max(len(v) for v in banks.values())
    return tot


def fwd_b_addr(hp, kg):  # pixel-major 64-B pixels, chunk swizzle
    return hp * 64 + ((kg ^ (((hp >> 2) & 1) << 1)) << 4)


worst = 0
for p0 in range(0, 64):
    addrs = [fwd_b_addr(p0 + (l & 15), l >> 4) for l in range(64)]
    worst = max(worst, cycles(addrs, B128_GROUPS, 16))
print("fwd B-frag ds_read_b128 (16x16x32), worst cycles over pixel offsets:", worst, "(ideal 4)")

# unswizzled for comparison
worst = 0
for p0 in range(0, 64):
    addrs = [(p0 + (l & 15)) * 64 + (l >> 4) * 16 for l in range(64)]
    worst = max(worst, cycles(addrs, B128_GROUPS, 16))
print("  same, no swizzle:", worst)


def wg_addr(pix, ch):  # pixel-major 128-B pixels (64 ch), 32-B block swizzle
    blk = (ch // 16) ^ ((pix >> 1) & 3)
    return pix * 128 + blk * 32 + (ch % 16) * 2


for stride in (1, 2):
    worst = 0
    for p0 in range(0, 64):
        for c0 in (0, 16, 32, 48):
            for rd in (0, 1):
                addrs = []
                for l in range(64):
                    g, q, p = l >> 4, (l & 15) >> 2, l & 3
                    pix = p0 + stride * (rd * 16 + 4 * g + q)
                    addrs.append(wg_addr(pix, c0 + 4 * p))
                worst = max(worst, cycles(addrs, HALVES, 8))
    print("wgrad tr-read b64 stride", stride, "worst cycles:", worst, "(ideal 2)")
