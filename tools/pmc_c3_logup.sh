# HBM traffic per kernel of the streamed configs[2] proof (separate FETCH_SIZE / WRITE_SIZE passes; 1 warm-up + 1 timed proof each)
# -> gpurun_out/r04_pmc_traffic_c3_logup.json
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c3_fetch /tmp/c3_write
( timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/c3_fetch -- python $R/tools/run_c3_logup.py 1 --no-constraints-only --no-segment ) > /dev/null 2> $R/gpurun_out/r04_pmc_c3_fetch.err
( timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/c3_write -- python $R/tools/run_c3_logup.py 1 --no-constraints-only --no-segment ) > /dev/null 2> $R/gpurun_out/r04_pmc_c3_write.err
cd $R
python tools/pmc_traffic_json.py /tmp/c3_fetch /tmp/c3_write 2 "C3 3731 cols x 2^22 rows with LogUp, streamed over 4 sub-cosets (bytes per PROOF; trace generation runs once: its kernels show half their bytes)" > gpurun_out/r04_pmc_traffic_c3_logup.json 2> gpurun_out/r04_pmc_traffic_c3.err
python - <<P
import json
t=json.load(open('gpurun_out/r04_pmc_traffic_c3_logup.json'))
for k,v in sorted(t['kernels'].items(), key=lambda kv:-(kv[1]['fetch_bytes_corrected']+kv[1]['write_bytes']))[:12]:
    print(k[:64], v['dispatches'], round(v['fetch_bytes_reported']/1e9,1), round(v['fetch_bytes_corrected']/1e9,1), round(v['write_bytes']/1e9,1))
P
