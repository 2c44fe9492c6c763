#!/bin/bash
# ONE script for every GPU call of a round (run through gpurun from the repo root):
#     bash tools/gpu_round.sh <tag> <step> [<step> ...]
# writes everything under gpurun_out/<tag>/ (copy what should be judged into profiles/).  Steps:
#   tests            the whole `pytest -m gpu` suite                      -> pytest.log
#   tests_reverse / tests_serialize   the suite in reverse order / under AMD_SERIALIZE_KERNEL=3 -> pytest_reverse.log / pytest_serialize.log
#   tests:<expr>     `pytest -m gpu -k <expr>`                            -> pytest_k.log
#   smoke            __graft_entry__.smoke()                              -> smoke.log
#   bench            bench.py with its defaults (all configs, PMC passes, CPU baseline)   -> bench.json
#   bench_driver     bench.py --gpus 1 --steps 20 --warmup 5 (the driver's flags)        -> bench_driver_flags.json
#   bench_mfcc       bench.py --op mfcc --no-configs                      -> bench_mfcc.json
#   prof             rocprofv3 --kernel-trace --stats of a headline run   -> prof_summary.txt
#   prof_configs     rocprofv3 --kernel-trace --stats of tools/bench_configs.py (every BASELINE config) -> prof_configs_summary.txt
#   configs          tools/bench_configs.py                               -> configs.jsonl
#   pmc:<op>:<kernel> rocprofv3 PMC passes (tools/pmc_groups_short.txt) of tools/run_op_once.py <op> -> pmc_<op>.txt
#   py:<file.py>     any tools/ script, stdout+stderr                     -> <file>.log
# Every step runs under its own `timeout`; a hung kernel cannot take the box with it.
set -x
R=$PWD
TAG=${1:-r4x}; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  case "$step" in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log ;;
    tests_reverse) timeout 1500 python -m pytest tests -m gpu -x -q --reverse-order > $O/pytest_reverse.log 2>&1; echo "pytest rc=$?" >> $O/pytest_reverse.log; tail -3 $O/pytest_reverse.log ;;
    tests_serialize) AMD_SERIALIZE_KERNEL=3 timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_serialize.log 2>&1; echo "pytest rc=$?" >> $O/pytest_serialize.log; tail -3 $O/pytest_serialize.log ;;
    tests:*) timeout 900 python -m pytest tests -m gpu -x -q -k "${step#tests:}" > $O/pytest_k.log 2>&1; echo "pytest rc=$?" >> $O/pytest_k.log; tail -15 $O/pytest_k.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log ;;
    bench) timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err ;;
    bench_driver) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; cat $O/bench_driver_flags.json | cut -c1-1500 ;;
    bench_mfcc) timeout 400 python bench.py --op mfcc --no-configs > $O/bench_mfcc.json 2> $O/bench_mfcc.err; cat $O/bench_mfcc.json ;;
    prof)
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o mel -- python $R/bench.py --steps 1000 --warmup 500 --no-cpu-baseline --no-traffic --no-configs > $R/$O/prof.log 2>&1)
      python tools/prof_summary.py $O/prof > $O/prof_summary.txt; head -8 $O/prof_summary.txt
      rm -rf $O/prof/*/ 2>/dev/null; find $O/prof -name "*.csv" -size +2M -delete 2>/dev/null ;;
    prof_configs)
      (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_configs -o cfg -- python $R/tools/bench_configs.py --steps 100 --warmup 50 > $R/$O/prof_configs.jsonl 2> $R/$O/prof_configs.err)
      python tools/prof_summary.py $O/prof_configs > $O/prof_configs_summary.txt; head -30 $O/prof_configs_summary.txt
      rm -rf $O/prof_configs/*/ 2>/dev/null; find $O/prof_configs -name "*.csv" -size +2M -delete 2>/dev/null ;;
    configs) timeout 600 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cut -c1-300 $O/configs.jsonl ;;
    pmc:*) a=${step#pmc:}; op=${a%%:*}; key=${a#*:}; timeout 900 bash tools/pmc_run.sh $O/pmc_$op tools/pmc_groups_short.txt "$key" python tools/run_op_once.py $op 4 > $O/pmc_$op.txt 2>&1; tail -40 $O/pmc_$op.txt; rm -rf $O/pmc_$op/p*/ 2>/dev/null ;;
    py:*) f=${step#py:}; b=$(basename ${f%% *} .py); timeout 900 python tools/$f > $O/$b.log 2>&1; echo "rc=$?" >> $O/$b.log; tail -40 $O/$b.log ;;
    *) echo "unknown step $step" ;;
  esac
done
