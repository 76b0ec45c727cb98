#!/bin/bash
# usage (GPU box): tools/pmc_wgrad_layers.sh -> gpurun_out/<ROUND>/wgrad_traffic_by_layer.txt: HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE, two --pmc passes) of every
# weight-gradient launch of one c2 backward pass, in launch order, beside the launch's algorithmic bytes (dz + layer input read once, one slab per workgroup written)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmcw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcw -o rd -- python3 $REPO/bench.py --steps 2 --warmup 1 --no_cpu_baseline --sustain_seconds 0 --prime_seconds 0 > $OUT/pmcw_rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcw -o wr -- python3 $REPO/bench.py --steps 2 --warmup 1 --no_cpu_baseline --sustain_seconds 0 --prime_seconds 0 > $OUT/pmcw_wr.log 2>&1
python3 - <<PY
import csv, glob, collections
def load(pat, name):
    rows = []
    for f in glob.glob("$OUT/pmcw/**/" + pat + "_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(rows)
rd, wr = load("rd", "FETCH_SIZE"), load("wr", "WRITE_SIZE")
def is_wg(k): return any(t in k for t in ("igemm_wgpp_kernel", "igemm_wgp64_kernel", "igemm_wgrad_kernel"))
def steps(rows):   # split the dispatch stream into steps at the colour adjust of the forward pass
    out, cur = [], []
    for d, k, v in rows:
        if "k_color_adjust" in k and "bwd" not in k:
            if cur: out.append(cur)
            cur = []
        cur.append((k, v))
    out.append(cur)
    return out
sr, sw = steps(rd), steps(wr)
# the first timed step of the default (two-stream) schedule: the one with 21 weight-gradient launches closest to the start
pick = lambda ss: next(s for s in ss[2:] if sum(1 for k, _ in s if is_wg(k)) == 21)
a, b = pick(sr), pick(sw)
wa = [(k, v) for k, v in a if is_wg(k) or "k_reduce_slabs" in k]
wb = [(k, v) for k, v in b if is_wg(k) or "k_reduce_slabs" in k]
lines = ["launch order of one backward pass (c2, B = 4): kernel | fetched MB (FETCH_SIZE KiB x 2) | written MB"]
tf = tw = 0.0
for (k, r), (_, w) in zip(wa, wb):
    name = "reduce" if "reduce" in k else ("wgpp" if "wgpp" in k else ("wgp64" if "wgp64" in k else "wgrad"))
    lines.append("  %-7s fetched %7.1f  written %7.1f" % (name, 2 * 1024 * r / 1e6, 1024 * w / 1e6))
    tf += 2 * 1024 * r / 1e6; tw += 1024 * w / 1e6
lines.append("  total fetched %.1f MB, written %.1f MB" % (tf, tw))
open("$OUT/wgrad_traffic_by_layer.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/pmcw
