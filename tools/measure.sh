#!/bin/bash
# usage (GPU box): tools/measure.sh  -> gpurun_out/${ROUND:-r06}/: the round's measured artefacts (copied to profiles/${ROUND:-r06}/ afterwards)
#   bench.json            python bench.py (default flags: two-stream timed region, roofline = chip time of that schedule, CPU baselines)
#   bench_c3.json / bench_c4.json   python bench.py --workload c3 / c4
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command in the DEFAULT two-stream schedule (tile shapes imported: no timing
#                         launches)
#   traffic.json          two --pmc passes (FETCH_SIZE / WRITE_SIZE) over the same command, stamped with the library's hash
#   hbm_kernels.md        achieved GB/s of the bandwidth-bound kernels (tools/hbm_kernels_report.py)
#   train_e2e.txt         tools/bench_train_e2e.py (the CLI's default --dropout 0.8, three input paths)
#   predict.json          sliding-window inference: c5 (604 px, stride 12) and the reference's published config (608 px, stride 110)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${ROUND:-r06}
mkdir -p $OUT $REPO/profiles/${ROUND:-r06}
export RSU_AUTOTUNE_FILE=$OUT/autotune.json
python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
unset RSU_AUTOTUNE_FILE
python3 $REPO/bench.py --workload c4 --no_cpu_baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python3 $REPO/bench.py --workload c3 --no_cpu_baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
export RSU_AUTOTUNE_FILE=$OUT/autotune.json
cd /tmp && export TMPDIR=/tmp
STEPS=10; WARM=2
rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python3 $REPO/bench.py --steps $STEPS --warmup $WARM --no_cpu_baseline --sustain_seconds 0 > $OUT/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o rd -- python3 $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline --sustain_seconds 0 > $OUT/log_rd.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o wr -- python3 $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline --sustain_seconds 0 > $OUT/log_wr.txt 2>&1
python3 - <<PY
import sqlite3, glob, csv, collections, json, hashlib
dbs = glob.glob("$OUT/prof/**/*.db", recursive=True)
nsteps = 0
if dbs:
    db = sqlite3.connect(dbs[0])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    nsteps = max([c for n, c, *_ in rows if "k_color_adjust" in n and "bwd" not in n] + [1])   # one colour adjust per forward pass
    with open("$OUT/kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([n, c, t, "%.1f" % a, "%.2f" % (100.0 * t / tot), mn, mx])
    fam = collections.OrderedDict()
    def family(n):
        for k, lab in (("igemm_wg_group", "igemm_wg_group (3x3 weight gradients, grouped)"), ("igemm_pp_kernel", "igemm_pp (3x3 fwd / bwd-data)"),
                       ("igemm_fwd2_kernel", "igemm_fwd2 (3x3: level-0 conv1, five-fragment shapes)"), ("igemm_wgpp", "igemm_wgpp"), ("igemm_wgp64", "igemm_wgp64"),
                       ("igemm_wgt_kernel", "igemm_wgt (convT weight gradients)"), ("igemm_wgrad_kernel", "igemm_wgrad (level-0 conv1 / ungrouped)"), ("igemm_ct", "igemm_ct (convT fwd / bwd-data)"),
                       ("k_reduce_slabs", "k_reduce_slabs*")):
            if k in n: return lab
        return n.split("(")[0][:40]
    for n, c, t, a, mn, mx in rows:
        e = fam.setdefault(family(n), [0, 0])
        e[0] += c; e[1] += t
    with open("$OUT/kernel_stats_per_step.md", "w") as f:
        f.write("Per-step view of kernel_stats.csv (rocprofv3 --kernel-trace --stats of \`python3 bench.py --steps $STEPS --warmup $WARM --no_cpu_baseline --sustain_seconds 0\`,\n"
                "the default two-stream schedule: in the backward pass a backward-data and a weight-gradient kernel run side by side on 128 CUs each, so their durations are those beside the other; %d steps in the trace incl. warm-up, priming and the three instrumented passes, of which two are single-stream)\n\n" % nsteps)
        f.write("| kernel family | launches / step | ms / step | avg us |\n|---|---|---|---|\n")
        tt = 0.0
        for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            f.write("| %s | %.1f | %.3f | %.1f |\n" % (k, c / nsteps, t / nsteps / 1e6, t / c / 1e3))
            tt += t / nsteps / 1e6
        f.write("| **total** | | **%.3f** | |\n" % tt)
    print(open("$OUT/kernel_stats_per_step.md").read())
def agg(pat, name):
    d = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc/**/" + pat + "_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d
rd, wr = agg("rd", "FETCH_SIZE"), agg("wr", "WRITE_SIZE")
def conv3(k):
    return "igemm_pp_kernel" in k or "igemm_wgpp_kernel" in k or "igemm_wgp64_kernel" in k or "igemm_wg_group" in k or ("igemm_fwd2_kernel" in k and ", 9, 3," in k) or ("igemm_wgrad_kernel" in k and ", 9, 3," in k) or "igemm_wg1_kernel" in k or "k_conv_first_fwd" in k
def wg(k):
    return "wgrad" in k or "wgpp" in k or "wgp64" in k or "wg_group" in k or "igemm_wg1" in k
fams = {"igemm_pp + igemm_fwd2 3x3 + k_conv_first_fwd (forward, backward-data)": lambda k: conv3(k) and not wg(k),
        "weight gradient launches (igemm_wgpp / igemm_wgp64 / igemm_wgrad per layer; igemm_wg_group in the grouped instrumented pass; level-0 conv1 = igemm_wg1)": lambda k: conv3(k) and wg(k),
        "k_reduce_slabs*": lambda k: "k_reduce_slabs" in k, "conv3x3 all": conv3}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over python3 bench.py --steps 3 --warmup 1, default two-stream schedule, "
                 "tile shapes of the bench run imported (tools/measure.sh); raw counter unit KiB; gfx950: FETCH_SIZE x 2 for 16-byte-per-lane reads",
       "workload": "num_layers=5 root_size=64 patch_size=388 batch 4",
       "lib_sha16": hashlib.sha256(open("$REPO/road_segmentation_unet_amd/librsu_hip.so", "rb").read()).hexdigest()[:16], "kernels": {}}
for fam, key in fams.items():
    r = [v for k, vs in rd.items() if key(k) for v in vs]
    w = [v for k, vs in wr.items() if key(k) for v in vs]
    if r and w:
        fb, wb = 2 * 1024 * sum(r) / len(r), 1024 * sum(w) / len(w)
        out["kernels"][fam] = {"launches_sampled": len(r), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                               "hbm_bytes_total_sampled": 2 * 1024 * sum(r) + 1024 * sum(w)}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
open("$OUT/nsteps.txt", "w").write(str(nsteps))
PY
python3 $REPO/tools/hbm_kernels_report.py $OUT/kernel_stats.csv $(cat $OUT/nsteps.txt) > $OUT/hbm_kernels.md; cat $OUT/hbm_kernels.md
rm -rf $OUT/prof $OUT/pmc
cd $REPO
# the contract line once more, now that traffic.json carries this library's hash (bench.py reads profiles/${ROUND:-r06}/traffic.json): this is bench.json
cp $OUT/traffic.json $REPO/profiles/${ROUND:-r06}/traffic.json
python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 tools/bench_predict.py --L 6 --dilated --images 1 --stride 12 --size 604 --batch 8 2>/dev/null | tail -1 > $OUT/predict_c5.txt
python3 tools/bench_predict.py --L 6 --dilated --images 10 --stride 110 --size 608 --batch 1 2>/dev/null | tail -1 > $OUT/predict_ref.txt
python3 - <<PY
import json, ast
c5 = ast.literal_eval(open("$OUT/predict_c5.txt").read())
rf = ast.literal_eval(open("$OUT/predict_ref.txt").read())
out = {"tool": "tools/bench_predict.py (ConvolutionalModel.predict on synthetic images, random-init weights, one MI355X, second call timed)",
       "c5_604px_stride12": {"config": "BASELINE.json configs[4]: num_layers=6 root_size=64 dilated, 604x604 image, stride 12, 6-way ensemble = 2166 tiles per image, shared-window path", **{k: v for k, v in c5.items() if k != "mask_shape"},
                             "seconds_per_image": c5["seconds"]},
       "reference_published_config_608px_stride110": {"config": "/root/reference/run.py:122-132: num_layers=6 dilated, patch 388, 608x608 test images, stride 110, 6-way ensemble = 54 tiles per image, batch_size 1; 10 images",
                             **{k: v for k, v in rf.items() if k != "mask_shape"}, "seconds_per_image": rf["seconds"] / 10.0,
                             "reference_published": "approximately 6 seconds per image of size 608x608 on one Nvidia GeForce Titan X (report/report.tex:254)"}}
json.dump(out, open("$OUT/predict.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
timeout 900 python3 tools/bench_train_e2e.py 16 2 2>/dev/null | tail -40 > $OUT/train_e2e.txt   # ConvolutionalModel.train() at the CLI's default --dropout 0.8, input path included
cut -c1-300 $OUT/bench.json
