#!/bin/bash
# PMC passes for the headline kernel (one rocprofv3 run per counter group; see MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  Usage (on the GPU box, from the repo root):
#   bash tools/pmc_mel.sh <outdir> [groups-file, one counter group per line]
R=$PWD
OUT=${1:-gpurun_out/pmc}
GROUPS_FILE=$(realpath ${2:-tools/pmc_groups_default.txt})
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/p$i -o pmc -- python $R/tools/run_mel_once.py 4 > $R/$OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done < $GROUPS_FILE
cd $R
python tools/pmc_summary.py $OUT melspec400
