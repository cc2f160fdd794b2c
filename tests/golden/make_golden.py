"""Generates the committed golden fixtures.

    python tests/golden/make_golden.py [--backend oracle|compressai|auto] [--out DIR] [input|kat|fp|hyperprior ...]

Backends (tests/golden/backends.py): `oracle` = the CPU restatement under oracle/, the only one that runs in the build
container (compressai / torchdistill / torchvision are not installed and not installable there); `compressai` = the
reference's real dependencies (CompressAI's entropy models, GDN layers, rANS coder, CDF quantiser; sc2bench's bottleneck
classes when the reference package is importable) on THE SAME seeded weights, writing THE SAME keys.  Every fixture records
which one produced it (`_provenance`); `tests/test_oracle_rans.py::test_fixture_provenance` fails when that and
`recipe.PINNED_BY` disagree.  With the oracle backend the vectors pin the oracle against regressions and give the GPU tests
fixed inputs / outputs; they are NOT outputs of a CompressAI binary (parity unpinned).  To upgrade the pin:

    pip install compressai && pip install -e <sc2-benchmark checkout>      # a machine with network access
    python tests/golden/make_golden.py --backend compressai                   # fails before writing anything if an import fails
    # set PINNED_BY = 'compressai' in tests/golden/recipe.py; pytest -m "not gpu" then checks the ORACLE against the reference

Only the fixtures travel (data); nothing of the reference does.
"""
import argparse
import json
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from backends import BackendUnavailable, get_backend  # noqa: E402
from recipe import fingerprint  # noqa: E402


def make_rans_kat(B, out_dir):
    kat = {'_provenance': ('self-derived from the restated algorithm (SURVEY.md 8(c)); C and pure-Python restatements agree; '
                           'not from a CompressAI binary') if B.name == 'oracle' else
           'compressai=={} (_CXX.pmf_to_quantized_cdf, ans.RansEncoder / RansDecoder)'.format(B.provenance()['compressai']),
           '_provenance_backend': B.name}
    pmf = [0.1, 0.2, 0.4, 0.2, 0.099, 0.001]
    cdf = B.pmf_to_quantized_cdf(pmf)
    kat['cdf_cases'] = [
        {'pmf': pmf, 'cdf': cdf},
        {'pmf': [1e-9, 0.5, 0.5 - 2e-9, 1e-9], 'cdf': B.pmf_to_quantized_cdf([1e-9, 0.5, 0.5 - 2e-9, 1e-9])},
        {'pmf': [0.25] * 4, 'cdf': B.pmf_to_quantized_cdf([0.25] * 4)},
        {'pmf': [1e-12] * 7 + [1.0], 'cdf': B.pmf_to_quantized_cdf([1e-12] * 7 + [1.0])},
    ]
    table = {'cdfs': [cdf], 'cdf_sizes': [7], 'offsets': [-2]}
    cases = []
    for syms in ([], [0, 1, -1, 2, -2, 0, 0, 1], [0, 3, -3, 40, -40, 1000, 0], [5] * 9, [-100000, 100000], [70000] * 3):
        idx = [0] * len(syms)
        enc = B.rans_encode(syms, idx, table['cdfs'], table['cdf_sizes'], table['offsets'])
        assert B.rans_decode(enc, idx, table['cdfs'], table['cdf_sizes'], table['offsets']) == syms
        cases.append({'symbols': syms, 'indexes': idx, 'hex': enc.hex()})
    # two-row table with different lengths and offsets, indexes alternate
    cdf2 = B.pmf_to_quantized_cdf([0.05, 0.9, 0.04, 0.01])
    width = max(len(cdf), len(cdf2))
    t2 = {'cdfs': [cdf + [0] * (width - len(cdf)), cdf2 + [0] * (width - len(cdf2))], 'cdf_sizes': [7, 5],
          'offsets': [-2, -1]}
    rng = random.Random(0)
    syms = [rng.randint(-6, 6) for _ in range(5000)]
    idx = [i % 2 for i in range(5000)]
    enc = B.rans_encode(syms, idx, t2['cdfs'], t2['cdf_sizes'], t2['offsets'])
    kat['table'] = table
    kat['cases'] = cases
    kat['table2'] = t2
    kat['case2'] = {'seed': 0, 'n': 5000, 'nbytes': len(enc),
                    'sha_prefix_hex': enc[:32].hex(), 'tail_hex': enc[-16:].hex()}
    with open(os.path.join(out_dir, 'rans_kat.json'), 'w') as f:
        json.dump(kat, f, indent=1)


def make_fp_golden(B, out_dir):
    m, x = B.fp_bottleneck()
    g = {'_provenance': B.provenance(), 'fingerprint': fingerprint(m), 'x': x}
    with torch.no_grad():
        latent = m.encoder(x)
        y_hat, lik = m.entropy_bottleneck(latent)
        g['latent'] = latent
        g['y_hat_eval'] = y_hat
        g['lik_eval'] = lik
        noise = torch.rand_like(latent) - 0.5
        y_noisy, lik_noisy = B.eb_with_noise(m.entropy_bottleneck, latent, noise)
        g['noise'] = noise
        g['y_hat_noise'] = y_noisy
        g['lik_noise'] = lik_noisy
        g['bits_eval'] = B.bpp_loss(y_hat, lik, 'sum')
        g['bpp_mean'] = B.bpp_loss(y_hat, lik, 'mean')
        g['bpp_batchmean'] = B.bpp_loss(y_hat, lik, 'batchmean')
        g['decoded'] = m.decoder(y_hat)
        g['aux_loss'] = m.aux_loss()
        g['gdn_in'] = torch.randn(2, 48, 7, 7)
        g['gdn_out'] = m.encoder[3](g['gdn_in'])
        g['igdn_out'] = B.gdn1(48, True)(g['gdn_in'])
        m.update()
        eb = m.entropy_bottleneck
        g['quantized_cdf'] = eb._quantized_cdf.clone()
        g['offset'] = eb._offset.clone()
        g['cdf_length'] = eb._cdf_length.clone()
        g['symbols'] = B.eb_symbols(eb, latent)
        enc = m.encode(x)
        g['strings_hex'] = [s.hex() for s in enc['strings'][0]]
        g['shape'] = list(enc['shape'])
        g['file_size_kb'] = B.file_size(enc)
        g['file_size_env'] = {'python': sys.version.split()[0], 'torch': torch.__version__}
        g['decoded_from_strings'] = m.decode(**enc)
    torch.save(g, os.path.join(out_dir, 'fp_golden.pt'))


def make_hyperprior_golden(B, out_dir):
    """Scale / mean-scale hyperprior bottlenecks (layer.py:553-817): Gaussian-conditional tables, likelihoods, indexes,
    both byte streams and the decoded output of the seeded oracle models."""
    import hashlib
    out = {'_provenance': B.provenance()}
    for name in ('SHPBasedResNetBottleneck', 'MSHPBasedResNetBottleneck'):
        m, x = B.hyperprior(name)
        g = {'fingerprint': fingerprint(m), 'x': x}
        with torch.no_grad():
            y = m.g_a(x)
            z = m.h_a(torch.abs(y) if name.startswith('SHP') else y)
            z_hat, z_lik = m.entropy_bottleneck(z)
            params = m.h_s(z_hat)
            scales, means = (params, None) if name.startswith('SHP') else params.chunk(2, 1)
            y_hat, y_lik = m.gaussian_conditional(y, scales, means=means)
            g.update(y=y, z=z, z_hat=z_hat, z_lik=z_lik, gaussian_params=params, y_hat=y_hat, y_lik=y_lik)
            noise = torch.rand_like(y) - 0.5
            g['noise_y'] = noise
            g['y_hat_noise'], g['y_lik_noise'] = B.gc_with_noise(m.gaussian_conditional, y, scales, means, noise)
            m.update()
            gc = m.gaussian_conditional
            g['gc_cdf_sha256'] = hashlib.sha256(gc._quantized_cdf.numpy().tobytes()).hexdigest()
            g['gc_cdf_shape'] = list(gc._quantized_cdf.shape)
            g['gc_offset'] = gc._offset.clone()
            g['gc_cdf_length'] = gc._cdf_length.clone()
            g['gc_cdf_rows'] = {i: gc._quantized_cdf[i, :int(gc._cdf_length[i])].clone() for i in (0, 1, 17, 63)}
            g['scale_table'] = gc.scale_table.clone()
            g['indexes'] = gc.build_indexes(scales)
            enc = m.encode(x)
            g['y_strings_hex'] = [s.hex() for s in enc['strings'][0]]
            g['z_strings_hex'] = [s.hex() for s in enc['strings'][1]]
            g['shape'] = list(enc['shape'])
            g['decoded'] = m.decode(**enc)
            g['file_size_kb'] = B.file_size(enc)
        out[name] = g
    torch.save(out, os.path.join(out_dir, 'hyperprior_golden.pt'))


def make_input_golden(B, out_dir):
    """the seeded factorized-prior codec (latent, tables, byte streams, reconstruction) and
    the codec feature-compression transform (PILTensorModule sizes / reconstruction digests, Pillow version recorded)."""
    import hashlib
    import PIL
    m, x = B.factorized_prior()
    g = {'_provenance': B.provenance(), 'fingerprint': fingerprint(m), 'x': x}
    with torch.no_grad():
        y = m.g_a(x)
        g['y'] = y
        g['gdn_in'] = torch.randn(2, 192, 5, 7)
        g['gdn_out'] = m.g_a[1](g['gdn_in'])
        g['igdn_out'] = m.g_s[1](g['gdn_in'])
        out = m(x)                                  # eval mode: dequantised latent, no noise
        g['x_hat_forward'] = out['x_hat']
        g['bits'] = float(-torch.log2(out['likelihoods']['y']).sum())
        m.update()
        eb = m.entropy_bottleneck
        g['cdf_sha256'] = hashlib.sha256(eb._quantized_cdf.numpy().tobytes()).hexdigest()
        g['cdf_shape'] = list(eb._quantized_cdf.shape)
        g['offset'], g['cdf_length'] = eb._offset.clone(), eb._cdf_length.clone()
        enc = m.compress(x)
        g['strings_hex'] = [s.hex() for s in enc['strings'][0]]
        g['shape'] = list(enc['shape'])
        g['file_size_kb'] = B.file_size(enc)
        g['x_hat'] = m.decompress(**enc)['x_hat']
    torch.manual_seed(3)
    feats = {'f512': torch.randn(512, 7, 7).abs() * 2 + 0.05, 'f5': torch.randn(5, 12, 10) + 3.0}
    g['pil'] = {'pillow': PIL.__version__, 'cases': {}}
    for name, t in feats.items():
        rec, size = B.pil_tensor_module(t, format='JPEG', quality=90)
        g['pil']['cases'][name] = {'x': t, 'file_size': size, 'recon_sha256': hashlib.sha256(rec.numpy().tobytes()).hexdigest()}
    g['pad'] = {'in_shape': [3, 224, 224], 'factor': 64, 'out_shape': list(B.adaptive_pad(torch.zeros(3, 224, 224), 64).shape)}
    torch.save(g, os.path.join(out_dir, 'input_golden.pt'))


MAKERS = {'input': make_input_golden, 'kat': make_rans_kat, 'fp': make_fp_golden, 'hyperprior': make_hyperprior_golden}


def main(argv=None):
    ap = argparse.ArgumentParser(description='(re)generate tests/golden/*')
    ap.add_argument('--backend', choices=['oracle', 'compressai', 'auto'], default='oracle')
    ap.add_argument('--out', default=HERE, help='directory the fixtures are written to (default: tests/golden)')
    ap.add_argument('which', nargs='*', choices=sorted(MAKERS) + [[]], help='fixtures to write (default: all)')
    args = ap.parse_args(argv)
    try:
        B = get_backend(args.backend)
        os.makedirs(args.out, exist_ok=True)
        # everything is produced in memory first by each maker and written at its end: a backend that turns out to lack a piece
        # (BackendUnavailable) stops the run, and the message says which fixtures had been written by then
        done = []
        for name in (args.which or ['input', 'kat', 'fp', 'hyperprior']):
            MAKERS[name](B, args.out)
            done.append(name)
    except BackendUnavailable as e:
        print('make_golden: {}'.format(e), file=sys.stderr)
        return 2
    print('golden fixtures {} written to {} by the {} backend'.format(done, args.out, B.name))
    return 0


if __name__ == '__main__':
    sys.exit(main())
