#!/bin/bash
# PMC passes over tools/ubench/mel400_lab (all LAB variants in one run).  bash tools/pmc_lab.sh <outdir>
R=$PWD
OUT=${1:-gpurun_out/pmc_lab}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/p$i -o pmc -- $R/tools/ubench/mel400_lab 2 > $R/$OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done <<'GRP'
SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
GRBM_GUI_ACTIVE
GRP
cd $R
python - $OUT <<'PY'
import collections, csv, glob, os, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "melspec400_kernel" not in k: continue
        lab = k.split("melspec400_kernel<")[1].split(">")[0].split(",")[0].strip() + "/" + k.split("melspec400_kernel<")[1].split(">")[0].split(",")[-1].strip()
        per[(lab, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (lab, disp, name), v in per.items():
        acc[lab][name].append(v)
names = sorted({n for lab in acc for n in acc[lab]})
labs = sorted(acc, key=lambda x: (int(x.split("/")[0]), x))
print("%-26s" % "counter (avg/dispatch)" + "".join("%14s" % ("LAB" + l) for l in labs))
for n in names:
    print("%-26s" % n + "".join("%14.3g" % (sum(acc[l][n][1:]) / max(1, len(acc[l][n][1:]))) if acc[l][n] else "%14s" % "-" for l in labs))
PY
