set -x
O=gpurun_out/r2w
mkdir -p $O
bash tools/pmc_run.sh $O/pmc_f16 tools/pmc_groups_rsm.txt resample python tools/run_op_once.py resample 3 > $O/pmc_f16.txt 2>&1
AAMD_RESAMPLE_FP32=1 bash tools/pmc_run.sh $O/pmc_f32 tools/pmc_groups_rsm.txt resample python tools/run_op_once.py resample 3 > $O/pmc_f32.txt 2>&1
tail -40 $O/pmc_f16.txt; tail -40 $O/pmc_f32.txt
rm -rf $O/pmc_f16/p*/ $O/pmc_f32/p*/ 2>/dev/null
