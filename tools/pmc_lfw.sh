#!/bin/bash
# PMC passes over lab variants of the biquad-cascade kernel (tools/lfw_ab.py).  bash tools/pmc_lfw.sh <outdir> NAME NAME ...
R=$PWD
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  i=0
  while read -r grp; do
    [ -z "$grp" ] && continue
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/${n}_p$i -o pmc -- python $R/tools/lfw_ab.py run $n --rounds 1 --launches 4 > $R/$OUT/${n}_p$i.log 2>&1 || echo "pass $i of $n failed: $grp"
  done <<'GRP'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
GRBM_GUI_ACTIVE
GRP
done
cd $R
python - $OUT "$@" <<'PY'
import collections, csv, glob, os, sys
d, names = sys.argv[1], sys.argv[2:]
acc = {n: collections.defaultdict(list) for n in names}
for n in names:
    for f in sorted(glob.glob(os.path.join(d, n + "_p*", "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "lfilter_wave_mover_kernel" not in r["Kernel_Name"]: continue
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (disp, c), v in per.items():
            acc[n][c].append(v)
cs = sorted({c for n in names for c in acc[n]})
print("%-26s" % "counter (avg/dispatch)" + "".join("%14s" % n for n in names))
for c in cs:
    print("%-26s" % c + "".join("%14.4g" % (sum(acc[n][c][2:]) / max(1, len(acc[n][c][2:]))) if acc[n][c] else "%14s" % "-" for n in names))
PY
