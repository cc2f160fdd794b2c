"""Per-kernel MFMA utilisation from the PMC passes of tools/pmc_mfma.sh.

Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles of busy matrix pipes summed
over SIMDs; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* count quad-cycles; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of
the cycles the dispatch was active (kernel cycles = / 8).  MFMA utilisation = MFMA_BUSY / (kernel cycles x 1024 SIMDs).
The FLOP count per launch is the algorithmic one, so `flop_per_busy_cycle` shows how close a busy
matrix pipe is to its 1024 FLOP / cycle / SIMD (bf16 16x16x32: 16384 FLOP in 16 cycles)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, 'g*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get('Kernel_Name') or ''
        res[name][row['Counter_Name']].append(float(row['Counter_Value']))
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, 'g1', '**', '*kernel_trace.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6)

N = 256
KERNELS = [   # (label, name pattern, algorithmic GFLOP per launch at bs 256)
    ('enc.conv0+gdn96', 'conv0_gdn96_kernel', 0.4118 * N), ('enc.conv2+gdn48', 'conv2_gdn48_kernel', 0.737 * N),
    ('dec.conv0+igdn512', 'conv2x2_gdn512_kernel', 1.9525 * N),
    ('dec.conv2+igdn256 (tile kernel)', 'Cfg8<256, 2, 4, true, 512, 2, 2', 3.5684 * N),
    ('dec.conv4 (tile kernel)', 'Cfg8<256, 2, 4, true, 256, 2, 2', 1.6442 * N),
    ('dec.conv2+igdn256 (win)', 'Geo2<55, 0, false>, 1,', 3.5684 * N), ('dec.conv2 (win)', 'Geo2<55, 0, false>, 0,', 3.1719 * N),
    ('dec.conv4 (win)', 'Geo2<56, 1, false>, 0,', 1.6442 * N), ('dec.conv4 + layer2.0 conv1 / downsample (win, tail)', 'Geo2<56, 1, false>, 2,', 2.0552 * N)]
if len(sys.argv) > 2:   # label:pattern:gflop triples replace the default list (tools/pmc_head.sh)
    KERNELS = [(a.split('|')[0], a.split('|')[1], float(a.split('|')[2])) for a in sys.argv[2:]]


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float('nan')


print('{:<30} {:>8} {:>9} {:>9} {:>9} {:>9} {:>10} {:>10}'.format('kernel', 'ms', 'mfma_util', 'wave_busy', 'wait_any', 'wait_inst', 'flop/busy', 'clock GHz'))
for label, pat, gflop in KERNELS:
    for name, d in res.items():
        if pat not in name:
            continue
        busy = med(d.get('SQ_VALU_MFMA_BUSY_CYCLES', []))
        gui = med(d.get('GRBM_GUI_ACTIVE', []))
        wave = med(d.get('SQ_WAVE_CYCLES', []))
        wait = med(d.get('SQ_WAIT_ANY', []))
        winst = med(d.get('SQ_WAIT_INST_ANY', []))
        active = med(d.get('SQ_ACTIVE_INST_ANY', []))
        ms = med(dur.get(name, []))
        cyc = gui / 8.0
        print('{:<30} {:>8.3f} {:>9.3f} {:>9.3f} {:>9.3f} {:>9.3f} {:>10.1f} {:>10.2f}'.format(
            label, ms, busy / (cyc * 1024.0), active / wave if wave else float('nan'), wait / wave if wave else float('nan'),
            winst / wave if wave else float('nan'), gflop * 1e9 / busy if busy else float('nan'), cyc / (ms * 1e6) if ms else float('nan')))
        print('    raw: MFMA_BUSY {:.3e}  MOPS_BF16 {:.3e}  INSTS_MFMA {:.3e}  SQ_BUSY {:.3e}  WAVE {:.3e}  WAIT_ANY {:.3e}  GUI_ACTIVE {:.3e}  LDS_CONFLICT {:.3e}  LDS_IDX_ACTIVE {:.3e}'.format(
            busy, med(d.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', [])), med(d.get('SQ_INSTS_MFMA', [])), med(d.get('SQ_BUSY_CYCLES', [])), wave, wait, gui,
            med(d.get('SQ_LDS_BANK_CONFLICT', [])), med(d.get('SQ_LDS_IDX_ACTIVE', []))))
