# Round-3 verification + profile collection (run through gpurun from the repo root):  bash tools/gpu_round3_final.sh <tag>
set -x
R=$PWD
O=gpurun_out/${1:-r3z}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o mel -- python $R/bench.py --steps 1000 --warmup 500 --no-cpu-baseline --no-traffic > $R/$O/prof.log 2>&1
cd $R
python tools/prof_summary.py $O/prof > $O/prof_summary.txt
tail -3 $O/pytest.log; tail -2 $O/smoke.log; cat $O/bench.json; cat $O/bench_driver_flags.json; head -8 $O/prof_summary.txt
timeout 400 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cat $O/configs.jsonl | cut -c1-260
rm -rf $O/prof/*/ 2>/dev/null; find $O/prof -name "*.csv" -size +2M -delete 2>/dev/null
