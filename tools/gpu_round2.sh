# Round-2 GPU session script (run through gpurun from the repo root):  bash tools/gpu_round2.sh <tag>
set -x
R=$PWD
O=gpurun_out/${1:-r2b}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 500 --warmup 100 --no-cpu-baseline --no-traffic > $O/bench_torchrun.json 2> $O/bench_torchrun.err
AAMD_NO_TORCH_SHIM=1 timeout 200 python tools/gpu_microbench.py mel > $O/micro_mel_ctypes.log 2>&1
timeout 200 python tools/gpu_microbench.py mel > $O/micro_mel_shim.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o mel -- python $R/bench.py --steps 1000 --warmup 500 --no-cpu-baseline --no-traffic > $R/$O/prof.log 2>&1
cd $R
python tools/prof_summary.py $O/prof > $O/prof_summary.txt
tail -3 $O/pytest.log; tail -2 $O/smoke.log; cat $O/bench.json; cat $O/bench_driver_flags.json; cat $O/micro_mel_*.log; head -20 $O/prof_summary.txt
