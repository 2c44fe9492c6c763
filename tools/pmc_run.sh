#!/bin/bash
# PMC passes for any command (one rocprofv3 run per counter group; MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   bash tools/pmc_run.sh <outdir> <groups-file> <kernel substring> <command...>     (from the repo root, on the GPU box)
R=$PWD
OUT=$1; GROUPS_FILE=$(realpath $2); KEY=$3; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  (cd $R && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/p$i -o pmc -- "$@" > $R/$OUT/p$i.log 2>&1) || echo "pass $i failed: $grp"
done < $GROUPS_FILE
cd $R
python tools/pmc_summary.py $OUT $KEY
