set -x
R=$PWD
mkdir -p gpurun_out/r1i
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r1i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1i/pytest.log
timeout 300 python bench.py > gpurun_out/r1i/bench.json 2> gpurun_out/r1i/bench.err
timeout 300 python tools/gpu_microbench.py mel spec mfcc resample lfilter fftconv > gpurun_out/r1i/micro.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1i/prof -o mel -- python $R/bench.py --steps 1000 --warmup 500 --no-cpu-baseline > $R/gpurun_out/r1i/prof.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/r1i/prof > gpurun_out/r1i/prof_summary.txt
printf 'FETCH_SIZE\nWRITE_SIZE\nGRBM_GUI_ACTIVE GRBM_COUNT\nSQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR\nSQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA\n' > /tmp/grp.txt
timeout 600 bash tools/pmc_mel.sh gpurun_out/r1i/pmc /tmp/grp.txt > gpurun_out/r1i/pmc_summary.txt 2>&1
tail -3 gpurun_out/r1i/pytest.log; cat gpurun_out/r1i/bench.json; cat gpurun_out/r1i/micro.log; cat gpurun_out/r1i/pmc_summary.txt
timeout 300 python tools/bench_configs.py > gpurun_out/r1i/configs.jsonl 2> gpurun_out/r1i/configs.err
timeout 200 python tools/bench_generic_shapes.py > gpurun_out/r1i/shapes.log 2>&1
timeout 200 python tools/gpu_microbench.py istft > gpurun_out/r1i/istft.log 2>&1
timeout 200 python tools/gpu_microbench.py rnnt > gpurun_out/r1i/rnnt.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r1i/smoke.log 2>&1
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 500 --warmup 100 --no-cpu-baseline > gpurun_out/r1i/bench_torchrun.json 2> gpurun_out/r1i/bench_torchrun.err
