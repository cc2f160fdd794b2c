"""Reads a rocprofv3 kernel trace (bench_kernel_trace.csv) of bench.py and reports, per hardware queue, how the time of the
steady part of the run splits into kernel time and gaps between consecutive kernels of that queue (dependent launches of one
HIP stream): the per-launch cost that the sum of kernel durations does not show."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5   # analyse the last `tail` fraction of the trace (the timed region)
t_all0 = min(int(r['Start_Timestamp']) for r in rows)
t_all1 = max(int(r['End_Timestamp']) for r in rows)
t_from = t_all1 - (t_all1 - t_all0) * tail
byq = defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp']) >= t_from:
        byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for q, ks in sorted(byq.items()):
    ks.sort()
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 100000]
    neg = [g for g in gaps if g < 0]
    big = [g for g in gaps if g >= 100000]
    print('queue {}: {} kernels, span {:.2f} ms, busy {:.2f} ms; gaps < 100 us: n {} sum {:.2f} ms mean {:.1f} us median {:.1f} us; '
          'gaps >= 100 us: n {} sum {:.2f} ms; overlapping: {}'.format(
              q, len(ks), span / 1e6, busy / 1e6, len(small), sum(small) / 1e6, (sum(small) / max(len(small), 1)) / 1e3,
              (sorted(small)[len(small) // 2] if small else 0) / 1e3, len(big), sum(big) / 1e6, len(neg)))
    names = defaultdict(lambda: [0, 0])
    for i in range(len(ks) - 1):
        g = ks[i + 1][0] - ks[i][1]
        if 0 <= g < 100000:
            n = ks[i + 1][2].split('(')[0][-60:]
            names[n][0] += 1
            names[n][1] += g
    for n, (c, g) in sorted(names.items(), key=lambda kv: -kv[1][1])[:8]:
        print('    gap before {:<62} n {:4d} mean {:6.1f} us'.format(n, c, g / c / 1e3))
