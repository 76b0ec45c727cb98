#!/bin/bash
# usage: tools/pmc_gen_ab.sh H Cin Cout [op=fwd]  -> SQ counters of one conv layer under igemm_fwd2 (RSU_FWD_GEN=2) and igemm_pp (RSU_FWD_GEN=4),
# shape 128x256 forced, tuning off; two rocprofv3 --pmc passes each (no trace domains beside --kernel-trace). Writes gpurun_out/pmc_gen_ab.csv
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
H=$1; CIN=$2; COUT=$3; OP=${4:-fwd}
cd /tmp && export TMPDIR=/tmp
export RSU_AUTOTUNE=0 RSU_FWD2_CFG=0
for G in ${GENS:-2 4}; do
  export RSU_FWD_GEN=$G
  OUT=$REPO/gpurun_out/pmc_gen$G
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -o p1 -- python3 $REPO/tools/one_layer.py $H $CIN $COUT $OP 4 30 > $OUT/log1.txt 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o p2 -- python3 $REPO/tools/one_layer.py $H $CIN $COUT $OP 4 30 > $OUT/log2.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
rows = []
for G in tuple(int(x) for x in "${GENS:-2 4}".split()):
    OUT = "$REPO/gpurun_out/pmc_gen%d" % G
    dur = collections.defaultdict(list)
    for f in glob.glob(OUT + "/**/p1_kernel_trace.csv", recursive=True) + glob.glob(OUT + "/p1_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(OUT + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "igemm" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        c = {n: sum(v) / len(v) for n, v in d.items()}
        us = sum(dur[k]) / max(1, len(dur[k]))
        mf = c.get("SQ_INSTS_MFMA", 0)
        ghz = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (us * 1e3) if us else 0
        rows.append([k.replace(",", " ")[:70], "%.1f" % us, int(mf), "%.2f" % ((c.get("SQ_INSTS_VALU", 0) - mf) / mf if mf else 0),
                     "%.2f" % (c.get("SQ_INSTS_SALU", 0) / mf if mf else 0), "%.2f" % (c.get("SQ_INSTS_LDS", 0) / mf if mf else 0),
                     "%.3f" % (c.get("SQ_INSTS_VMEM", 0) / mf if mf else 0),
                     "%.3f" % (c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0),
                     "%.2f" % ghz,
                     "%.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * us * 1e3 * ghz) if us and ghz else 0),
                     "%.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0)])
hdr = "kernel,us_under_pmc,MFMA_insts,nonMFMA_VALU_per_MFMA,SALU_per_MFMA,LDS_insts_per_MFMA,VMEM_insts_per_MFMA,WAIT_ANY_frac_of_wave_cycles,clock_GHz_GRBM,mfma_pipe_util_at_that_clock,lds_bank_conflict_per_active_cycle"
open("$REPO/gpurun_out/pmc_gen_ab.csv", "w").write(hdr + "\n" + "\n".join(",".join(map(str, r)) for r in rows) + "\n")
print(hdr)
for r in rows:
    print(",".join(map(str, r)))
PY
