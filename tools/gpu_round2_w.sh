# kernel-trace stats over every BASELINE config + HBM traffic counters of the fftconvolve delay-line kernel (separate PMC passes)
R=$PWD
O=gpurun_out/r2w
mkdir -p $O
cat > /tmp/fdl_once.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import torch, audio_amd.functional as F
x = torch.rand(32, 8, 480000, device="cuda") - 0.5
h = torch.randn(1, 1, 24000, device="cuda") * 0.01
for _ in range(6):
    F.fftconvolve(x, h)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfg -o cfg -- python $R/tools/bench_configs.py > $R/$O/cfg.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o pmc -- python /tmp/fdl_once.py $R > $R/$O/pmc_$c.log 2>&1
done
cd $R
python tools/prof_summary.py $O/cfg > $O/cfg_summary.txt 2>/dev/null; head -30 $O/cfg_summary.txt
python - $O <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"][:60], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, _), v in per.items():
            acc[k].append(v)
    for k, v in acc.items():
        if "overlap_save" in k or "spectrum" in k:
            print(c, k, "avg KB per dispatch: %.0f over %d" % (sum(v) / len(v), len(v)))
PY
rm -rf $O/cfg/*/ $O/pmc_*/*/ 2>/dev/null; find $O -name "*.csv" -size +1M -delete 2>/dev/null
