# Round 6: HBM traffic per kernel of the configs[2] proof ON HEAD — trace handed over, 2 sub-cosets, the query pass
# storing its terms (no atomics) — from separate FETCH_SIZE / WRITE_SIZE passes (1 warm-up + 1 timed proof each), every pass behind its
# own timeout; then round 5's atomic form under the same FETCH_SIZE pass (the one that did not return in round 5).
# -> gpurun_out/r06_pmc_traffic_c3_logup.json, gpurun_out/r06_pmc_c3_atomic_form.txt
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c3_fetch /tmp/c3_write /tmp/c3_fetch_atomic
SECONDS=0
( timeout -k 10 ${TMO:-420} rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/c3_fetch -- python $R/tools/run_c3_logup.py 1 --no-constraints-only --no-segment ) > $R/gpurun_out/r06_pmc_c3_fetch.out 2> $R/gpurun_out/r06_pmc_c3_fetch.err
echo "FETCH_SIZE pass (terms stored): rc=$? ${SECONDS}s"
SECONDS=0
( timeout -k 10 ${TMO:-420} rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/c3_write -- python $R/tools/run_c3_logup.py 1 --no-constraints-only --no-segment ) > $R/gpurun_out/r06_pmc_c3_write.out 2> $R/gpurun_out/r06_pmc_c3_write.err
echo "WRITE_SIZE pass (terms stored): rc=$? ${SECONDS}s"
cd $R
python tools/pmc_traffic_json.py /tmp/c3_fetch /tmp/c3_write 2 "C3 3731 cols x 2^22 rows with LogUp on round 6's HEAD: trace handed over, 2 sub-cosets, query pass with stored terms (bytes per PROOF; trace generation runs once: its kernels show half their bytes)" > gpurun_out/r06_pmc_traffic_c3_logup.json 2> gpurun_out/r06_pmc_traffic_c3.err
python - <<P
import json
t=json.load(open('gpurun_out/r06_pmc_traffic_c3_logup.json'))
for k,v in sorted(t['kernels'].items(), key=lambda kv:-(kv[1]['fetch_bytes_corrected']+kv[1]['write_bytes']))[:14]:
    print(k[:64], v['dispatches'], round(v['fetch_bytes_reported']/1e9,1), round(v['fetch_bytes_corrected']/1e9,1), round(v['write_bytes']/1e9,1))
P
if [ "${ATOMIC_PASS:-1}" = 1 ]; then
cd /tmp; SECONDS=0
( POWDR_QUERY_SELECT=2 timeout -k 10 ${TMO:-420} rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/c3_fetch_atomic -- python $R/tools/run_c3_logup.py 1 --no-constraints-only --no-segment ) > $R/gpurun_out/r06_pmc_c3_fetch_atomic.out 2> $R/gpurun_out/r06_pmc_c3_fetch_atomic.err
echo "FETCH_SIZE pass, POWDR_QUERY_SELECT=2 (64-bit atomic sums, round 5's form): rc=$? ${SECONDS}s (124 = killed by the timeout)" | tee $R/gpurun_out/r06_pmc_c3_atomic_form.txt
grep -h "prove_ms\|verify_rc" $R/gpurun_out/r06_pmc_c3_fetch_atomic.out | head -4 | tee -a $R/gpurun_out/r06_pmc_c3_atomic_form.txt
fi
