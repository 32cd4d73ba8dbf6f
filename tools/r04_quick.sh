# a short check on the GPU: selected tests + one default bench line
cd $GRAFT_REPO_ROOT
T=${GLAMR_TAG:-r04x}
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "${GLAMR_K:-smpl or e2e}" 2>&1 | grep -E "passed|failed|error" | tail -4 > gpurun_out/${T}_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_tests.log; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/${T}_bench.json') if l.startswith('{"metric"')][0])
print('value', round(j['value']), 'ms_per_step', round(j['ms_per_step'],2), 'host stream', round(j.get('host_inclusive_sequences_per_sec',0)), 'single', round(j.get('host_inclusive_single_call_sequences_per_sec',0)), 'first', round(j.get('host_inclusive_first_call_on_a_new_stream_sequences_per_sec',0)))
print({k:v for k,v in j.get('pipeline',{}).items()})
PY
