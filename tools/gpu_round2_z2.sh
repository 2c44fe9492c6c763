# rocprof kernel stats over the per-config benchmark + PMC of the lfilter mover kernel
set -x
R=$PWD
O=gpurun_out/r2z2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o cfg -- python $R/tools/bench_configs.py --steps 40 --warmup 20 > $R/$O/prof_cfg.log 2>&1
cd $R
python tools/prof_summary.py $O/prof > $O/configs_kernel_stats.txt; head -30 $O/configs_kernel_stats.txt
bash tools/pmc_run.sh $O/pmc_lfw tools/pmc_groups_sq.txt lfilter python tools/run_op_once.py lfilter 3 > $O/pmc_lfilter_mover.txt 2>&1; tail -30 $O/pmc_lfilter_mover.txt
rm -rf $O/prof/*/ $O/pmc_lfw/p*/ 2>/dev/null; find $O -name "*.csv" -size +1M -delete 2>/dev/null
