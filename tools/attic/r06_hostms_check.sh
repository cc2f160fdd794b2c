export TMPDIR=/tmp
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_benchpath.py -q -x 2>&1 | tail -2
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-bs1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), d['config']['host_coder_steps']['steps'], d['config']['host_coder_steps'].get('host_ms_per_batch_measured'), d['config'].get('coder_group_plan'))"; done
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-bs1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), d['config']['host_coder_steps']['steps'], d['config']['host_coder_steps'].get('host_ms_per_batch_measured'), d['config'].get('coder_group_plan'))"
