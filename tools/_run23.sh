mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r02_pytest23.log
LEGS="--no-cpu-baseline --no-logup-leg --no-segment-leg --no-callmajor-leg --no-copy-ceiling"
timeout 300 python bench.py --steps 5 --warmup 2 $LEGS > gpurun_out/r02_bench_scaled.json 2>/dev/null
python - <<P
import json
d=json.load(open('gpurun_out/r02_bench_scaled.json')); s=d['stage_ms']; print(d['value']/1e9, d['ms_per_step'], s['leaf_hash_kernel'], s['compress_kernel'], s['compress_tail_kernel'], s['lde_fused_kernel'])
P
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_pmc_valu
( timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r02_pmc_valu -- python $R/bench.py --log-height 18 --steps 1 --warmup 1 --no-cpu-baseline --no-logup-leg --no-copy-ceiling --no-segment-leg --no-callmajor-leg ) > $R/gpurun_out/r02_pmc_valu.log 2>&1
cd $R
python tools/pmc_valu_json.py gpurun_out/r02_pmc_valu profiles/r02_microbench_opcodes.txt 18 2022 2 > gpurun_out/r02_valu_model.json 2> gpurun_out/r02_valu_model.err
rm -rf gpurun_out/r02_pmc_valu
head -c 300 gpurun_out/r02_valu_model.json; echo
tail -3 gpurun_out/r02_pytest23.log
