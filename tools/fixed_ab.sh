#!/bin/bash
# usage (GPU box): tools/fixed_ab.sh name "libA.so libB.so ..." ["H,Cin,Cout,cfg ..."] [reps] -> gpurun_out/${ROUND:-r06}/fixed_<name>.txt: forward / backward-data
# times of single layers at FIXED tile shapes (tools/pp_fixed.py) under several builds of the library, alternating on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT; cd $REPO
SPECS=${3:-"570,64,64,1 570,64,64,5 282,128,128,0 282,128,128,4 138,256,256,0 392,128,64,1 390,64,64,5 198,128,128,0 66,512,512,0"}
REPS=${4:-3}
for rep in $(seq 1 $REPS); do
  for lib in $2; do
    echo "$lib: $(RSU_LIB_PATH=$REPO/$lib timeout 300 python3 tools/pp_fixed.py $SPECS 2>/dev/null | tail -1)"
  done
done | tee $OUT/fixed_$1.txt
python3 - $OUT/fixed_$1.txt <<'PY'
import sys, re, collections
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    lib, rest = line.split(": ", 1)
    for part in rest.strip().split(" | "):
        m = re.match(r"(\S+) fwd ([\d.]+) bwd ([\d.]+)", part)
        rows.setdefault(lib, collections.OrderedDict()).setdefault(m.group(1), []).append((float(m.group(2)), float(m.group(3))))
libs = list(rows)
base = rows[libs[0]]
print("medians (us), relative to %s" % libs[0])
for lib in libs:
    out = []
    tf = tb = bf = bb = 0.0
    for spec, v in rows[lib].items():
        f = sorted(x[0] for x in v)[len(v) // 2]; b = sorted(x[1] for x in v)[len(v) // 2]
        f0 = sorted(x[0] for x in base[spec])[len(base[spec]) // 2]; b0 = sorted(x[1] for x in base[spec])[len(base[spec]) // 2]
        out.append("%s f %.1f(%+.1f%%) b %.1f(%+.1f%%)" % (spec, f, 100 * (f / f0 - 1), b, 100 * (b / b0 - 1)))
        tf += f; tb += b; bf += f0; bb += b0
    print("%s: SUM fwd %.1f (%+.1f%%) bwd %.1f (%+.1f%%) | %s" % (lib, tf, 100 * (tf / bf - 1), tb, 100 * (tb / bb - 1), " | ".join(out)))
PY
