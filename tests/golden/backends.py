"""Two producers of the golden fixtures behind one interface.

`OracleBackend`      the CPU restatement under oracle/ -- what runs in the build container, where CompressAI, torchdistill and
                     torchvision are neither installed nor installable.  Fixtures it writes say `_provenance.backend = 'oracle'`
                     and pin the oracle against regressions only (parity unpinned).
`CompressaiBackend`  the REAL third-party arithmetic: `compressai` (EntropyBottleneck, GaussianConditional, GDN / GDN1,
                     bmshj2018_factorized, `ans.RansEncoder`, `_CXX.pmf_to_quantized_cdf`) and, when the reference package is
                     importable as `sc2bench` (`pip install -e <reference checkout>`: setup.py:28 pulls compressai>=1.2.3), its
                     own classes (`sc2bench.models.layer.{FP,SHP,MSHP}BasedResNetBottleneck`, `sc2bench.transforms.codec.
                     PILTensorModule`, `sc2bench.transforms.misc.AdaptivePad`).  Fixtures it writes say
                     `_provenance.backend = 'compressai'` with the package versions: on a machine where it runs,
                     `python tests/golden/make_golden.py --backend compressai` upgrades every fixture to a pin on the reference's
                     own output, and `pytest -m "not gpu"` then checks the ORACLE against it.

Both backends produce THE SAME seeded models: weights never come from the backend's own constructors' random draws -- the
oracle model is built from the seeded recipe (recipe.py) and its state dict is loaded into the reference class (the oracle
mirrors the reference's state-dict keys: that is part of what the fixtures pin).  Reference modules are constructed under
`torch.random.fork_rng()` so that the recipe's later draws (inputs, noise) are the same tensors under either backend.

Nothing here travels to the GPU box except as data: the fixtures.  This file is test infrastructure; the product package never
imports it.

STATUS of CompressaiBackend: written against the CompressAI 1.2.x API as sc2bench calls it (sc2bench/models/layer.py:2-6,
360-398, 506-547, 627-719) and NOT RUN in the build container (the import fails there, which `make_golden.py --backend
compressai` reports before touching any fixture).
"""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _version(mod_name):
    try:
        from importlib.metadata import version
        return version(mod_name)
    except Exception:
        mod = sys.modules.get(mod_name)
        return getattr(mod, '__version__', 'unknown')


class OracleBackend(object):
    name = 'oracle'

    def __init__(self):
        from oracle import cpu_ref as R
        from oracle import cpu_ref_input as RI
        from oracle import rans, rans_py
        self.R, self.RI, self.rans, self.rans_py = R, RI, rans, rans_py

    def provenance(self):
        return {'backend': 'oracle',
                'what': 'oracle/ (CPU restatement of the reference path): pins the oracle against regressions; NOT output of a '
                        'CompressAI binary -- parity unpinned',
                'torch': torch.__version__, 'python': sys.version.split()[0]}

    # ---- models: (module, input) of the seeded recipes
    def fp_bottleneck(self):
        from recipe import build_oracle_bottleneck
        return build_oracle_bottleneck(self.R)

    def hyperprior(self, name):
        from recipe import build_oracle_hyperprior
        return build_oracle_hyperprior(self.R, name)

    def factorized_prior(self):
        from recipe import build_oracle_factorized_prior
        return build_oracle_factorized_prior(self.RI, self.R)

    def gdn1(self, channels, inverse):
        return self.R.GDN1(channels, inverse=inverse)

    # ---- entropy models with a GIVEN noise tensor (the reference draws its own inside quantize('noise'))
    def eb_with_noise(self, eb, x, noise):
        return eb(x, training=True, noise=noise)

    def gc_with_noise(self, gc, y, scales, means, noise):
        return gc(y, scales, means=means, training=True, noise=noise)

    def eb_symbols(self, eb, x):
        return eb.symbols(x)

    # ---- scalars / objects
    def bpp_loss(self, y_hat, lik, reduction):
        return self.R.bpp_loss(y_hat, lik, reduction)

    def file_size(self, obj):
        return self.R.file_size(obj)

    def pil_tensor_module(self, t, **save_kwargs):
        return self.RI.pil_tensor_module(t, **save_kwargs)

    def adaptive_pad(self, t, factor):
        return self.RI.adaptive_pad(t, factor=factor)

    # ---- the integer coder
    def pmf_to_quantized_cdf(self, pmf, precision=16):
        cdf = [int(v) for v in self.rans.pmf_to_quantized_cdf(pmf, precision)]
        assert cdf == self.rans_py.pmf_to_quantized_cdf(pmf, precision), 'C and pure-Python restatements disagree'
        return cdf

    def rans_encode(self, symbols, indexes, cdfs, cdf_sizes, offsets):
        enc = self.rans.encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets)
        assert enc == self.rans_py.encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets), 'C and pure-Python restatements disagree'
        return enc

    def rans_decode(self, stream, indexes, cdfs, cdf_sizes, offsets):
        return [int(v) for v in self.rans.decode_with_indexes(stream, indexes, cdfs, cdf_sizes, offsets)]


class BackendUnavailable(RuntimeError):
    pass


class CompressaiBackend(OracleBackend):
    """The same interface on the reference's real dependencies.  Weights come from the oracle's seeded recipe through the state
    dict; every number written to a fixture comes out of CompressAI / sc2bench code."""
    name = 'compressai'

    def __init__(self):
        super().__init__()      # (the oracle builds the seeded weights; none of its arithmetic reaches a fixture)
        try:
            self.compressai = importlib.import_module('compressai')
            self.em = importlib.import_module('compressai.entropy_models')
            self.layers = importlib.import_module('compressai.layers')
            self.ans = importlib.import_module('compressai.ans')
            self.cxx = importlib.import_module('compressai._CXX')
        except ImportError as e:
            raise BackendUnavailable(
                'the compressai backend needs `compressai>=1.2.3` (the reference\'s setup.py:28): {!r}.  Nothing was written.  In this '
                'build container the package is neither installed nor installable (no network): run this command on a machine where '
                '`pip install compressai` (and, for the sc2bench classes, `pip install -e <sc2-benchmark checkout>`) works, then '
                'commit the regenerated fixtures and set PINNED_BY = "compressai" in tests/golden/recipe.py.'.format(e))
        try:
            self.sc2 = importlib.import_module('sc2bench.models.layer')
            self.sc2_codec = importlib.import_module('sc2bench.transforms.codec')
            self.sc2_misc = importlib.import_module('sc2bench.transforms.misc')
            self.sc2_analysis = importlib.import_module('sc2bench.analysis')
        except ImportError:
            self.sc2 = self.sc2_codec = self.sc2_misc = self.sc2_analysis = None

    def provenance(self):
        return {'backend': 'compressai', 'compressai': _version('compressai'),
                'sc2bench': _version('sc2bench') if self.sc2 is not None else 'not importable: bottleneck classes composed from compressai modules by the oracle\'s recipe',
                'torch': torch.__version__, 'python': sys.version.split()[0],
                'what': 'outputs of the reference\'s own dependencies on the seeded recipe weights: a pin'}

    # ---- helpers
    def _transplant(self, make_ref, oracle_module):
        """reference module with the oracle module's (seeded) parameters and buffers; RNG untouched."""
        with torch.random.fork_rng():
            ref = make_ref()
        sd = {k: v.clone() for k, v in oracle_module.state_dict().items()}
        missing, unexpected = ref.load_state_dict(sd, strict=False) if not hasattr(ref, 'entropy_bottleneck') else self._load_compression_model(ref, sd)
        # every learnable tensor must have been carried over: a silent leftover would make the fixture describe OTHER weights
        learn = {k for k, _ in ref.named_parameters()}
        assert not (learn & set(missing)), 'state-dict keys the oracle does not mirror: {}'.format(sorted(learn & set(missing)))
        assert not unexpected, 'oracle keys the reference does not have: {}'.format(sorted(unexpected))
        ref.eval()
        return ref

    @staticmethod
    def _load_compression_model(ref, sd):
        # CompressionModel.load_state_dict resizes the CDF buffers from the checkpoint first (compressai/models/base.py); the
        # oracle's un-updated models carry them empty, as the reference's do
        try:
            res = ref.load_state_dict(sd, strict=False)
        except TypeError:
            res = ref.load_state_dict(sd)
        return (list(getattr(res, 'missing_keys', [])), list(getattr(res, 'unexpected_keys', []))) if res is not None else ([], [])

    def _need_sc2(self):
        if self.sc2 is None:
            raise BackendUnavailable('the bottleneck fixtures need the reference package importable as `sc2bench` '
                                     '(pip install -e <sc2-benchmark checkout>); compressai alone was found.  Nothing was written.')

    # ---- models
    def fp_bottleneck(self):
        self._need_sc2()
        om, x = super().fp_bottleneck()
        return self._transplant(lambda: self.sc2.FPBasedResNetBottleneck(), om), x

    def hyperprior(self, name):
        self._need_sc2()
        om, x = super().hyperprior(name)
        return self._transplant(lambda: getattr(self.sc2, name)(), om), x

    def factorized_prior(self):
        zoo = importlib.import_module('compressai.zoo')
        om, x = super().factorized_prior()
        return self._transplant(lambda: zoo.bmshj2018_factorized(quality=8, metric='mse', pretrained=False), om), x

    def gdn1(self, channels, inverse):
        return self.layers.GDN1(channels, inverse=inverse)

    # ---- entropy models with a given noise tensor: the reference's own likelihood code on (x + noise)
    @staticmethod
    def _first(v):
        return v[0] if isinstance(v, tuple) else v      # (_likelihood returns (likelihood, lower, upper) from 1.2.4 on)

    def eb_with_noise(self, eb, x, noise):
        out = x + noise
        C = out.shape[1]
        perm = [1, 0] + list(range(2, out.dim()))
        values = out.permute(*perm).contiguous()
        shape = values.size()
        lik = self._first(eb._likelihood(values.reshape(C, 1, -1)))
        if eb.use_likelihood_bound:
            lik = eb.likelihood_lower_bound(lik)
        inv = [perm.index(i) for i in range(len(perm))]
        return out, lik.reshape(shape).permute(*inv).contiguous()

    def gc_with_noise(self, gc, y, scales, means, noise):
        out = y + noise
        lik = self._first(gc._likelihood(out, scales, means))
        if gc.use_likelihood_bound:
            lik = gc.likelihood_lower_bound(lik)
        return out, lik

    def eb_symbols(self, eb, x):
        medians = eb._get_medians().detach()
        spatial = x.dim() - 2
        medians = eb._extend_ndims(medians, spatial).expand(x.size(0), *([-1] * (spatial + 1)))
        return eb.quantize(x, 'symbols', medians)

    # ---- scalars / objects
    def bpp_loss(self, y_hat, lik, reduction):
        # the reference's own class (sc2bench/loss.py:6-37; a torchdistill mid-level loss reading the entropy module's hooked output
        # from the student's io dict)
        self._need_sc2()
        loss_mod = importlib.import_module('sc2bench.loss')
        return loss_mod.BppLoss('entropy_bottleneck', reduction=reduction)({'entropy_bottleneck': {'output': (y_hat, lik)}})

    def file_size(self, obj):
        import pickle
        return sys.getsizeof(pickle.dumps(obj)) / 1024       # torchdistill file_util.get_binary_object_size (sc2bench/analysis.py:133)

    def pil_tensor_module(self, t, **save_kwargs):
        if self.sc2_codec is None:
            raise BackendUnavailable('PILTensorModule needs the reference package importable as `sc2bench`')
        return self.sc2_codec.PILTensorModule(returns_file_size=True, **save_kwargs)(t)

    def adaptive_pad(self, t, factor):
        if self.sc2_misc is None:
            raise BackendUnavailable('AdaptivePad needs the reference package importable as `sc2bench`')
        return self.sc2_misc.AdaptivePad(factor=factor)(t)

    # ---- the integer coder
    def pmf_to_quantized_cdf(self, pmf, precision=16):
        return [int(v) for v in self.cxx.pmf_to_quantized_cdf([float(p) for p in pmf], precision)]

    def rans_encode(self, symbols, indexes, cdfs, cdf_sizes, offsets):
        cdfs = [[int(v) for v in row] for row in cdfs]
        return self.ans.RansEncoder().encode_with_indexes([int(s) for s in symbols], [int(i) for i in indexes], cdfs,
                                                          [int(v) for v in cdf_sizes], [int(v) for v in offsets])

    def rans_decode(self, stream, indexes, cdfs, cdf_sizes, offsets):
        cdfs = [[int(v) for v in row] for row in cdfs]
        return [int(v) for v in self.ans.RansDecoder().decode_with_indexes(stream, [int(i) for i in indexes], cdfs,
                                                                           [int(v) for v in cdf_sizes], [int(v) for v in offsets])]


def get_backend(name):
    """'oracle' | 'compressai' | 'auto' (compressai when it imports, else the oracle with a message on stderr)."""
    if name == 'oracle':
        return OracleBackend()
    if name == 'compressai':
        return CompressaiBackend()
    try:
        return CompressaiBackend()
    except BackendUnavailable as e:
        print('make_golden: {}\nmake_golden: falling back to the ORACLE backend (fixtures will say so in _provenance).'.format(e), file=sys.stderr)
        return OracleBackend()
