#!/bin/bash
# usage (GPU box): tools/r03_pmc.sh -> gpurun_out/r03/pmc_step_kernels.csv: SQ counters of the MFMA kernels of one config-2 step in the
# single-stream schedule (each kernel alone on the chip; three --pmc passes with --kernel-trace only), per kernel name averaged over launches
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03; mkdir -p $OUT/pmcq
cd /tmp && export TMPDIR=/tmp
export RSU_WGRAD_STREAM=0
CMD="python3 $REPO/bench.py --steps 2 --warmup 1 --no_cpu_baseline --sustain_seconds 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmcq -o p1 -- $CMD > $OUT/pmcq/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/pmcq -o p2 -- $CMD > $OUT/pmcq/log2.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmcq -o p3 -- $CMD > $OUT/pmcq/log3.txt 2>&1
python3 - <<PY
import csv, glob, collections
def short(n):
    return n.split("(")[0].replace("void ", "")[:70]
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/pmcq/**/p3_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmcq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "igemm" not in k and "conv_first" not in k: continue
        cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_step_kernels.csv", "w") as f:
    f.write("kernel,launches,us_under_pmc,MFMA_insts,nonMFMA_VALU_per_MFMA,SALU_per_MFMA,LDS_insts_per_MFMA,VMEM_insts_per_MFMA,WAIT_ANY_frac_of_wave_cycles,LDS_bank_conflict_frac,clock_GHz_GRBM,mfma_pipe_util_at_that_clock\n")
    for k, d in sorted(cnt.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
        a = {c: sum(v) / len(v) for c, v in d.items()}
        if "SQ_INSTS_MFMA" not in a or not dur.get(k): continue
        us = sum(dur[k]) / len(dur[k])
        m = a["SQ_INSTS_MFMA"]
        clk = a.get("GRBM_GUI_ACTIVE", 0) / 8.0 / (us * 1e-6) / 1e9 if us > 0 else 0
        util = m * 16.0 / (1024.0 * us * 1e-6 * clk * 1e9) if clk > 0 else 0
        f.write("%s,%d,%.1f,%d,%.2f,%.2f,%.2f,%.3f,%.3f,%.4f,%.2f,%.3f\n" % (k.replace(",", " "), len(dur[k]), us, m, (a["SQ_INSTS_VALU"] - m) / m, a["SQ_INSTS_SALU"] / m,
                a["SQ_INSTS_LDS"] / m, a.get("SQ_INSTS_VMEM", 0) / m, a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_LDS_BANK_CONFLICT"] / max(1.0, a["SQ_LDS_IDX_ACTIVE"]), clk, util))
print(open("$OUT/pmc_step_kernels.csv").read())
PY
rm -rf $OUT/pmcq
