"""Prints the fields of a bench.py JSON line that matter during tuning."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, b = d['roofline'], d['bottleneck_forward']
print('K={} value={:.0f} img/s  ms/step={:.3f}  roofline.frac={:.3f} ({} {:.3f} ms)  bottleneck {:.3f} ms frac {:.3f}  match={}'.format(
    d['steps'], d['value'], d['ms_per_step'], r['frac'], r['kernel'], r['kernel_ms'], b.get('ms_per_batch_sum_of_its_launches', b.get('ms_per_batch_sum_of_mfma_kernels')),
    b['frac_of_mfma_peak'], d.get('bitstream_match')))
print('  kernels_ms', json.dumps(d.get('kernels_ms')))
if 'stand_alone' in b:
    print('  stand-alone {:.3f} ms frac {:.3f}'.format(b['stand_alone']['ms_per_batch'], b['stand_alone']['frac_of_mfma_peak']))
if 'rans' in d:
    print('  rans enc {:.2f} dec {:.2f} ms'.format(d['rans']['encode_ms'], d['rans']['decode_ms']))
if 'f32_mode' in d:
    print('  f32_mode {:.0f} img/s'.format(d['f32_mode']['images_per_s']))
