"""-m gpu: the bottleneck at the shapes of BASELINE configs 4 and 5: 513 x 513 (PASCAL VOC crop,
odd width) and 800 x 1216 (a typical Faster R-CNN batch) -- encoder, range coder and decoder against the oracle, the
lossless round trip, and the device coder byte-equal to the oracle coder on the device latent."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-12)).item()


def _pair(S, R, dev, seed):
    torch.manual_seed(seed)
    ref = R.FPBasedResNetBottleneck()
    R.perturb_quantiles(ref.entropy_bottleneck)
    with torch.no_grad():
        ref.encoder[4].weight.mul_(30.0)
        for g in (ref.encoder[1], ref.encoder[3], ref.decoder[1], ref.decoder[3]):
            g.gamma.add_(0.02 * torch.rand_like(g.gamma))
    ref.eval()
    m = S.FPBasedResNetBottleneck()
    m.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    m.eval().to(dev)
    m.update()
    ref.update(force=True)
    return m, ref


@pytest.mark.parametrize('N,H,W', [(2, 513, 513), (1, 800, 1216), (1, 333, 500)])
def test_bottleneck_at_detection_and_segmentation_shapes(S, R, dev, N, H, W):
    torch.set_num_threads(32)
    m, ref = _pair(S, R, dev, H)
    x = torch.rand(N, 3, H, W)
    oh = (((H + 4 - 5) // 2 + 1) + 4 - 5) // 2 + 1 - 1
    ow = (((W + 4 - 5) // 2 + 1) + 4 - 5) // 2 + 1 - 1
    with torch.no_grad():
        latent = m.analysis(x.to(dev))
        assert latent.shape == (N, 24, oh, ow)
        ref_latent = ref.encoder(x)
        assert rel(latent, ref_latent) < 1.5e-2          # bf16 operands through 5 layers vs the f32 oracle
        # integer path, bit-exact on the device latent; lossless round trip
        enc = m.encode(x.to(dev))
        assert tuple(enc['shape']) == (oh, ow) and len(enc['strings'][0]) == N
        assert enc['strings'][0] == ref.entropy_bottleneck.compress(latent.cpu())
        y_hat = m.entropy_bottleneck.decompress(enc['strings'][0], enc['shape'])
        y_q, _ = m.entropy_bottleneck(latent)
        assert torch.equal(y_hat, y_q)
        out = m.decode(**enc)
        assert out.shape == (N, 256, oh + 1, ow + 1)
        assert rel(out, ref.decode(**enc)) < 1.5e-2       # the same bytes through the oracle's decoder
        # the three modes agree (noise is the only stochastic part; eval uses rounding)
        assert torch.equal(m(x.to(dev)), out)
