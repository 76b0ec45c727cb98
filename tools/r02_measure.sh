#!/bin/bash
# usage (GPU box): tools/r02_measure.sh  -> gpurun_out/r02/{bench.json, kernel_stats.csv, traffic.json, autotune.json}
# 1. bench.py (default flags) -- also exports the tile shapes it measured; 2. rocprofv3 --kernel-trace --stats of the same command with
# that table imported (no timing launches: call counts = launches per step x steps); 3. two --pmc passes (FETCH_SIZE / WRITE_SIZE
# cannot share one, MI355X_MICROARCH.md "HBM"; FETCH_SIZE x 2 for 16-byte-per-lane reads on gfx950) -> HBM bytes per conv launch,
# stamped with the sha-256 of the librsu_hip.so that ran.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02
rm -rf $OUT; mkdir -p $OUT
export RSU_AUTOTUNE_FILE=$OUT/autotune.json
python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
# the profiled runs keep every launch on ONE stream (RSU_WGRAD_STREAM=0): in the default schedule a backward-data and a weight-gradient
# kernel share the chip and a trace would charge each with the time it spent beside the other. bench.py's roofline figure comes from
# the same single-stream plan (its instrumented pass), so the per-kernel averages of the two agree.
export RSU_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r02 -- python3 $REPO/bench.py --steps 10 --warmup 2 --no_cpu_baseline --sustain_seconds 0 > $OUT/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o rd -- python3 $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline --sustain_seconds 0 > $OUT/log_rd.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o wr -- python3 $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline --sustain_seconds 0 > $OUT/log_wr.txt 2>&1
python3 - <<PY
import sqlite3, glob, csv, collections, json, hashlib
dbs = glob.glob("$OUT/prof/**/*.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open("$OUT/kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([n, c, t, "%.1f" % a, "%.2f" % (100.0 * t / tot), mn, mx])
    print("kernels:", len(rows), "total ms:", tot / 1e6)
def agg(pat, name):
    d = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc/**/" + pat + "_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d
rd, wr = agg("rd", "FETCH_SIZE"), agg("wr", "WRITE_SIZE")
def conv3(k):
    return "igemm_pp_kernel" in k or "igemm_wgpp_kernel" in k or "igemm_wgp64_kernel" in k or ("igemm_fwd2_kernel" in k and ", 9, 3," in k) or ("igemm_wgrad_kernel" in k and ", 9, 3," in k)
def wg(k):
    return "wgrad" in k or "wgpp" in k or "wgp64" in k
fams = {"igemm_pp + igemm_fwd2 3x3 (forward, backward-data)": lambda k: conv3(k) and not wg(k),
        "igemm_wgpp + igemm_wgp64 + igemm_wgrad 3x3 (weight gradient)": lambda k: conv3(k) and wg(k), "conv3x3 all": conv3}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over python3 bench.py --steps 3 --warmup 1 with the "
                 "tile shapes of the bench run imported (tools/r02_measure.sh); raw counter unit KiB; gfx950: FETCH_SIZE x 2 for 16-byte-per-lane reads",
       "workload": "num_layers=5 root_size=64 patch_size=388 batch 4",
       "lib_sha16": hashlib.sha256(open("$REPO/road_segmentation_unet_amd/librsu_hip.so", "rb").read()).hexdigest()[:16], "kernels": {}}
for fam, key in fams.items():
    r = [v for k, vs in rd.items() if key(k) for v in vs]
    w = [v for k, vs in wr.items() if key(k) for v in vs]
    if r and w:
        fb, wb = 2 * 1024 * sum(r) / len(r), 1024 * sum(w) / len(w)
        out["kernels"][fam] = {"launches_sampled": len(r), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
cut -c1-400 $OUT/bench.json
