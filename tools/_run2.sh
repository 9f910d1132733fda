mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r02_pytest2.log
( timeout 120 ./tools/microbench_opcodes ) > gpurun_out/r02_microbench_opcodes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_valu -- python $GRAFT_REPO_ROOT/bench.py --log-height 18 --steps 1 --warmup 1 --no-cpu-baseline --no-logup-leg --no-copy-ceiling ) > $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_valu.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_valu_json.py gpurun_out/r02_pmc_valu gpurun_out/r02_microbench_opcodes.txt 18 2022 2 > gpurun_out/r02_valu_model.json 2> gpurun_out/r02_valu_model.err
find gpurun_out/r02_pmc_valu -name "*.csv" | head; du -sh gpurun_out/r02_pmc_valu
tail -5 gpurun_out/r02_pytest2.log; cat gpurun_out/r02_valu_model.json | head -8; tail -3 gpurun_out/r02_valu_model.err
