"""A/B on the GPU box: stage-1 training step with / without a host-policy switch (default `train_fused_conv2`; argv[1] names another),
bench.py --mode train, alternating."""
import json
import subprocess
import sys

SWITCH = sys.argv[1] if len(sys.argv) > 1 else 'train_fused_conv2'
VALUES = {True: sys.argv[2], False: sys.argv[3]} if len(sys.argv) > 3 else {True: 'True', False: 'False'}   # (e.g. wgrad_ct 0 128)
res = {True: [], False: []}
for rep in range(3):
    for fused in (True, False):
        code = ("import sys; sys.argv = ['bench.py', '--mode', 'train', '--steps', '20', '--warmup', '5'];"
                "import importlib; S = importlib.import_module('sc2bench_amd');"
                "S.hip.configure({}={}); import runpy; runpy.run_path('bench.py', run_name='__main__')".format(SWITCH, VALUES[fused]))
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print(out.stdout[-2000:], out.stderr[-2000:])
            continue
        d = json.loads(line[-1])
        res[fused].append(d['ms_per_step'])
        print('{}={}'.format(SWITCH, VALUES[fused]), round(d['ms_per_step'], 3), 'ms/step', round(d['value']), 'images/s', flush=True)
for k, v in res.items():
    if v:
        print(SWITCH, '=', k, ': min', round(min(v), 3), 'median', round(sorted(v)[len(v) // 2], 3), 'ms per 256-image step')
