#!/bin/bash
# usage: tools/pmc_traffic.sh <tag>   -> HBM-side traffic of the conv MFMA kernels during bench.py (two separate --pmc passes, as
# MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass; gfx950: FETCH_SIZE x 2 for 16-B/lane reads)
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o rd -- python3 $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/log_rd.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o wr -- python3 $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/log_wr.txt 2>&1
python3 - <<PY
import csv, glob, collections, json
def agg(pat, name):
    d = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + pat + "_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d
rd, wr = agg("rd", "FETCH_SIZE"), agg("wr", "WRITE_SIZE")
out = {}
for fam, key in (("igemm_fwd2 3x3", lambda k: "igemm_fwd2_kernel" in k and ", 9, 3," in k), ("igemm_wgrad 3x3", lambda k: "igemm_wgrad_kernel" in k and ", 9, 3," in k)):
    r = [v for k, vs in rd.items() if key(k) for v in vs]
    w = [v for k, vs in wr.items() if key(k) for v in vs]
    out[fam] = {"launches": len(r), "FETCH_SIZE_raw_avg": sum(r) / max(1, len(r)), "WRITE_SIZE_raw_avg": sum(w) / max(1, len(w))}
allr = [v for k, vs in rd.items() if "igemm" in k and ", 9, 3," in k for v in vs]
allw = [v for k, vs in wr.items() if "igemm" in k and ", 9, 3," in k for v in vs]
out["conv3x3 all"] = {"launches": len(allr), "FETCH_SIZE_raw_avg": sum(allr) / max(1, len(allr)), "WRITE_SIZE_raw_avg": sum(allw) / max(1, len(allw))}
json.dump(out, open("$OUT/traffic_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -1 $OUT/log_rd.txt | cut -c1-120
