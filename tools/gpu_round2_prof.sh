# Round-2 profile collection (run through gpurun from the repo root):  bash tools/gpu_round2_prof.sh <tag>
set -x
R=$PWD
O=gpurun_out/${1:-r2j}
mkdir -p $O
timeout 400 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o mel -- python $R/bench.py --steps 1000 --warmup 500 --no-cpu-baseline --no-traffic > $R/$O/prof.log 2>&1
cd $R
python tools/prof_summary.py $O/prof > $O/prof_summary.txt
printf 'FETCH_SIZE\nWRITE_SIZE\nGRBM_GUI_ACTIVE GRBM_COUNT\nSQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR\nSQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA\nSQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL\n' > /tmp/grp.txt
timeout 600 bash tools/pmc_mel.sh $O/pmc /tmp/grp.txt > $O/pmc_summary.txt 2>&1
cat $O/configs.jsonl; head -12 $O/prof_summary.txt; cat $O/pmc_summary.txt
