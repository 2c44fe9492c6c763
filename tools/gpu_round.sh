set -x
R=$PWD
mkdir -p gpurun_out/r1h
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r1h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1h/pytest.log
timeout 300 python bench.py > gpurun_out/r1h/bench.json 2> gpurun_out/r1h/bench.err
timeout 300 python tools/gpu_microbench.py mel spec mfcc resample lfilter fftconv > gpurun_out/r1h/micro.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1h/prof -o mel -- python $R/bench.py --steps 1000 --warmup 500 --no-cpu-baseline > $R/gpurun_out/r1h/prof.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/r1h/prof > gpurun_out/r1h/prof_summary.txt
printf 'FETCH_SIZE\nWRITE_SIZE\nGRBM_GUI_ACTIVE GRBM_COUNT\nSQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR\nSQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA\n' > /tmp/grp.txt
timeout 600 bash tools/pmc_mel.sh gpurun_out/r1h/pmc /tmp/grp.txt > gpurun_out/r1h/pmc_summary.txt 2>&1
tail -3 gpurun_out/r1h/pytest.log; cat gpurun_out/r1h/bench.json; cat gpurun_out/r1h/micro.log; cat gpurun_out/r1h/pmc_summary.txt
timeout 300 python tools/bench_configs.py > gpurun_out/r1h/configs.jsonl 2> gpurun_out/r1h/configs.err
timeout 200 python tools/bench_generic_shapes.py > gpurun_out/r1h/shapes.log 2>&1
timeout 200 python tools/gpu_microbench.py istft > gpurun_out/r1h/istft.log 2>&1
