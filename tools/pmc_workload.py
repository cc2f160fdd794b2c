"""HBM traffic of the bottleneck forward inside a `bench.py --workload <name>` step, from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only -- tools/pmc_workload.sh makes them).  Per kernel: median over its
launches of FETCH_SIZE x 2 (gfx950: 128-byte requests are tallied at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, in bytes; the
bottleneck forward = the sum over its launches of one step.  Usage: python tools/pmc_workload.py <dir> <workload> [json to update]"""
import csv, glob, json, os, sys
from collections import defaultdict

BOTTLENECK = ('conv0_gdn96_kernel', 'conv2_gdn48_kernel', 'conv5s2_patch_kernel', 'conv2x2_c48_kernel', 'conv2x2_gdn512_kernel',
              'conv2x2_win_kernel', '<128, 32, 4, 1, true, 48, 2, 2', '<128, 48, 4, 1, true, 96, 5, 5')
out, workload = sys.argv[1], sys.argv[2]
res = defaultdict(lambda: defaultdict(list))
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(os.path.join(out, 'pmc_' + counter, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name') or row.get('Kernel Name') or ''
            if (row.get('Counter_Name') or row.get('Counter Name')) == counter and any(q in name for q in BOTTLENECK):
                res[name][counter].append(float(row.get('Counter_Value') or row.get('Counter Value') or 0))
per_kernel, total = {}, 0.0
for name, d in res.items():
    f, w = sorted(d.get('FETCH_SIZE', [0])), sorted(d.get('WRITE_SIZE', [0]))
    b = f[len(f) // 2] * 1024 * 2 + w[len(w) // 2] * 1024
    short = name.replace('void ', '').replace('(anonymous namespace)::', '')[:70]
    per_kernel[short] = {'hbm_bytes_per_launch': b, 'launches_sampled': len(f)}
    total += b
    print('{:<72} {:9.1f} MB (n={})'.format(short, b / 1e6, len(f)))
print('bottleneck forward of one {} step: {:.1f} MB'.format(workload, total / 1e6))
if len(sys.argv) > 3:
    path = sys.argv[3]
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[workload] = {'hbm_bytes_per_bottleneck_forward': total, 'per_kernel': per_kernel,
                       'source': 'rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, separate passes over bench.py --workload ' + workload}
    json.dump(table, open(path, 'w'), indent=1)
