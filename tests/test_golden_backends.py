"""The golden fixtures' provenance, and the generator's two backends (tests/golden/backends.py).

* every committed fixture says which backend wrote it, and that is the one tests/golden/recipe.py expects (`PINNED_BY`);
* `make_golden.py --backend compressai` stops with a clear message, before writing anything, where CompressAI does not import;
* the CompressAI adapter's plumbing (weight transplant through the state dict, the given-noise likelihood paths, symbols,
  coder / CDF calls, PILTensorModule / AdaptivePad wrappers) is exercised against a STAND-IN `compressai` / `sc2bench` package
  assembled from the oracle's own classes: through the adapter it must write fixtures value-identical to the committed ones.
  That proves the adapter asks the right questions of the API sc2bench uses -- not that CompressAI gives the oracle's answers
  (nothing in this container can: parity unpinned)."""
import json
import os
import subprocess
import sys
import types

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')
sys.path.insert(0, GOLDEN)
import recipe  # noqa: E402

PT = ('fp_golden.pt', 'hyperprior_golden.pt', 'input_golden.pt')


def _same(a, b, path=''):
    if isinstance(a, dict):
        ka, kb = set(a) - {'_provenance', '_provenance_backend'}, set(b) - {'_provenance', '_provenance_backend'}
        assert ka == kb, (path, ka ^ kb)
        for k in ka:
            _same(a[k], b[k], '{}/{}'.format(path, k))
    elif isinstance(a, torch.Tensor):
        assert a.dtype == b.dtype and a.shape == b.shape, path
        if a.is_floating_point():
            # (the same torch CPU ops on the same weights: equal up to how many threads the reductions were split over -- an earlier
            #  test of the suite may have changed torch's thread count; integer tensors, byte streams and table hashes stay exact)
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), path
        else:
            assert torch.equal(a, b), path
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (u, v) in enumerate(zip(a, b)):
            _same(u, v, '{}/{}'.format(path, i))
    elif isinstance(a, float) and isinstance(b, float):
        # (fingerprints are f64 sums over a tensor: their last bits depend on how torch splits the reduction over threads)
        assert a == b or abs(a - b) <= 1e-9 * max(abs(a), abs(b)), (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_fixture_provenance():
    for name in PT:
        prov = torch.load(os.path.join(GOLDEN, name), weights_only=False)['_provenance']
        assert prov['backend'] == recipe.PINNED_BY, '{} was written by the {} backend, recipe.PINNED_BY says {}'.format(name, prov['backend'], recipe.PINNED_BY)
        if prov['backend'] == 'compressai':
            assert prov['compressai'] not in ('', 'unknown')
    kat = json.load(open(os.path.join(GOLDEN, 'rans_kat.json')))
    assert kat['_provenance_backend'] == recipe.PINNED_BY
    assert ('compressai==' in kat['_provenance']) == (recipe.PINNED_BY == 'compressai')


def test_compressai_backend_refuses_cleanly_without_the_package(tmp_path):
    try:
        import compressai  # noqa: F401
        pytest.skip('compressai imports here: regenerate the fixtures with it (tests/golden/make_golden.py --backend compressai)')
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, 'make_golden.py'), '--backend', 'compressai', '--out', str(tmp_path)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'Nothing was written' in r.stderr and 'compressai>=1.2.3' in r.stderr
    assert os.listdir(str(tmp_path)) == []
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, 'make_golden.py'), '--backend', 'auto', '--out', str(tmp_path), 'kat'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'falling back to the ORACLE backend' in r.stderr
    assert json.load(open(os.path.join(str(tmp_path), 'rans_kat.json')))['_provenance_backend'] == 'oracle'


def _stand_in_packages():
    """`compressai` / `sc2bench` module objects whose classes are the oracle's (same names, same call signatures as the reference's
    dependencies where the adapter touches them)."""
    from oracle import cpu_ref as R
    from oracle import cpu_ref_input as RI
    from oracle import rans

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m

    class RansEncoder(object):
        def encode_with_indexes(self, symbols, indexes, cdfs, cdf_lengths, offsets):
            assert all(isinstance(v, int) for v in symbols) and isinstance(cdfs[0], list)      # pybind11 takes Python ints / lists
            return rans.encode_with_indexes(symbols, indexes, cdfs, cdf_lengths, offsets)

    class RansDecoder(object):
        def decode_with_indexes(self, stream, indexes, cdfs, cdf_lengths, offsets):
            return [int(v) for v in rans.decode_with_indexes(stream, indexes, cdfs, cdf_lengths, offsets)]

    def pmf_to_quantized_cdf(pmf, precision):
        assert isinstance(pmf, list) and all(isinstance(p, float) for p in pmf)
        return [int(v) for v in rans.pmf_to_quantized_cdf(pmf, precision)]

    class PILTensorModule(torch.nn.Module):
        def __init__(self, returns_file_size=False, open_kwargs=None, **save_kwargs):
            super().__init__()
            self.returns_file_size, self.open_kwargs, self.save_kwargs = returns_file_size, open_kwargs, save_kwargs

        def forward(self, x):
            rec, size = RI.pil_tensor_module(x, open_kwargs=self.open_kwargs, **self.save_kwargs)
            return (rec, size) if self.returns_file_size else rec

    class AdaptivePad(torch.nn.Module):
        def __init__(self, fill=0, padding_position='hw', padding_mode='constant', factor=128):
            super().__init__()
            self.kw = dict(fill=fill, padding_position=padding_position, padding_mode=padding_mode, factor=factor)

        def forward(self, x):
            return RI.adaptive_pad(x, **self.kw)

    def bmshj2018_factorized(quality, metric='mse', pretrained=False):
        assert not pretrained
        return RI.bmshj2018_factorized(quality, metric)

    class BppLoss(torch.nn.Module):
        def __init__(self, entropy_module_path, reduction='mean'):
            super().__init__()
            self.entropy_module_path, self.reduction = entropy_module_path, reduction

        def forward(self, student_io_dict, *args, **kwargs):
            return R.bpp_loss(*student_io_dict[self.entropy_module_path]['output'], self.reduction)

    mods = {
        'compressai': mod('compressai', __version__='0.0-stand-in'),
        'compressai.entropy_models': mod('compressai.entropy_models', EntropyBottleneck=R.EntropyBottleneck, GaussianConditional=R.GaussianConditional),
        'compressai.layers': mod('compressai.layers', GDN1=R.GDN1, GDN=RI.GDN),
        'compressai.ans': mod('compressai.ans', RansEncoder=RansEncoder, RansDecoder=RansDecoder),
        'compressai._CXX': mod('compressai._CXX', pmf_to_quantized_cdf=pmf_to_quantized_cdf),
        'compressai.zoo': mod('compressai.zoo', bmshj2018_factorized=bmshj2018_factorized),
        'sc2bench': mod('sc2bench', __version__='0.0-stand-in'),
        'sc2bench.models': mod('sc2bench.models'),
        'sc2bench.models.layer': mod('sc2bench.models.layer', FPBasedResNetBottleneck=R.FPBasedResNetBottleneck,
                                     SHPBasedResNetBottleneck=R.SHPBasedResNetBottleneck, MSHPBasedResNetBottleneck=R.MSHPBasedResNetBottleneck),
        'sc2bench.transforms': mod('sc2bench.transforms'),
        'sc2bench.transforms.codec': mod('sc2bench.transforms.codec', PILTensorModule=PILTensorModule),
        'sc2bench.transforms.misc': mod('sc2bench.transforms.misc', AdaptivePad=AdaptivePad),
        'sc2bench.analysis': mod('sc2bench.analysis'),
        'sc2bench.loss': mod('sc2bench.loss', BppLoss=BppLoss),
    }
    return mods


def test_compressai_adapter_on_a_stand_in_package_writes_the_committed_fixtures(tmp_path, monkeypatch):
    try:
        import compressai  # noqa: F401
        pytest.skip('compressai imports here: the real backend is the test')
    except ImportError:
        pass
    for name, m in _stand_in_packages().items():
        monkeypatch.setitem(sys.modules, name, m)
    import backends
    import make_golden
    B = backends.CompressaiBackend()
    assert B.name == 'compressai' and B.provenance()['backend'] == 'compressai' and B.sc2 is not None
    # the transplant really builds a SECOND module and carries every learnable tensor over
    om, _ = backends.OracleBackend().fp_bottleneck()
    rm, _ = B.fp_bottleneck()
    assert rm is not om and all(torch.equal(a, b) for a, b in zip(om.state_dict().values(), rm.state_dict().values()))
    for which in ('input', 'kat', 'fp', 'hyperprior'):
        make_golden.MAKERS[which](B, str(tmp_path))
    for name in PT:
        new, old = torch.load(os.path.join(str(tmp_path), name), weights_only=False), torch.load(os.path.join(GOLDEN, name), weights_only=False)
        assert new['_provenance']['backend'] == 'compressai'
        _same(new, old, name)
    new, old = json.load(open(os.path.join(str(tmp_path), 'rans_kat.json'))), json.load(open(os.path.join(GOLDEN, 'rans_kat.json')))
    assert new['_provenance_backend'] == 'compressai' and 'compressai==' in new['_provenance']
    _same(new, old, 'kat')
