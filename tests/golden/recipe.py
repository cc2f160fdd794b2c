"""Deterministic recipe for the golden bottleneck (shared by make_golden.py and the tests).

The 1.3 M weights are NOT stored in the fixture (5 MB); they are rebuilt from torch.manual_seed(0) on the
CPU generator and checked against the stored fingerprint.
"""
import torch


def build_oracle_bottleneck(R):
    torch.manual_seed(0)
    m = R.FPBasedResNetBottleneck()
    R.perturb_quantiles(m.entropy_bottleneck)
    with torch.no_grad():
        m.encoder[4].weight.mul_(40.0)   # spreads the latent over several quantisation bins
        for g in (m.encoder[1], m.encoder[3], m.decoder[1], m.decoder[3]):
            g.gamma.add_(0.02 * torch.rand_like(g.gamma))   # non-diagonal gamma
        for f in m.entropy_bottleneck.factors:
            f.add_(0.3 * torch.randn_like(f))               # exercise the tanh gates
    m.eval()
    x = torch.rand(2, 3, 32, 32)
    return m, x


def fingerprint(module):
    return {k: float(v.double().abs().sum()) for k, v in module.state_dict().items() if v.numel() > 0}


def build_oracle_hyperprior(R, name):
    """Seeded SHP / MSHP oracle bottleneck (name = class name) with non-trivial tables, and its input."""
    torch.manual_seed(1)
    m = getattr(R, name)()
    R.perturb_quantiles(m.entropy_bottleneck)
    with torch.no_grad():
        m.g_a[4].weight.mul_(40.0)              # spreads the latent over several quantisation bins
        m.h_a[2].weight.mul_(12.0)              # ... and the hyper-latent
        m.h_s[4].weight.mul_(6.0)               # scales / means well away from the 0.11 floor
        for g in (m.g_a[1], m.g_a[3], m.g_s[1], m.g_s[3]):
            g.gamma.add_(0.02 * torch.rand_like(g.gamma))
    m.eval()
    x = torch.rand(2, 3, 32, 32)
    return m, x


def build_oracle_factorized_prior(RI, R, quality=8):
    """Seeded bmshj2018-factorized oracle model (quality 8: N = 192, M = 320, the BASELINE config) with non-trivial
    GDN matrices, biases, tables and latent spread, and its input (already a multiple of 64 on both sides)."""
    torch.manual_seed(2)
    m = RI.bmshj2018_factorized(quality)
    eb = m.entropy_bottleneck
    with torch.no_grad():
        C = eb.channels
        q = torch.zeros(C, 1, 3)
        for c in range(C):
            q[c, 0, 0], q[c, 0, 1], q[c, 0, 2] = -(3 + c % 5), 0.25 * (c % 3), 4 + c % 7
        eb.quantiles.copy_(q)
        for mod in list(m.g_a) + list(m.g_s):
            if isinstance(mod, RI.GDN):
                mod.gamma.add_(0.01 * torch.rand_like(mod.gamma))
            elif mod.bias is not None:
                mod.bias.add_(0.05 * torch.randn_like(mod.bias))
        m.g_a[6].weight.mul_(60.0)      # spreads the latent over several quantisation bins
    m.eval()
    x = torch.rand(2, 3, 64, 128)
    return m, x


# Which backend of tests/golden/make_golden.py the COMMITTED fixtures are expected to come from: 'oracle' (the build container:
# parity unpinned) or 'compressai' (regenerated where the reference's dependencies run: a pin).  Regenerating with the other
# backend without changing this line -- or changing this line without regenerating -- fails tests/test_golden_backends.py.
PINNED_BY = 'oracle'
