"""HIP-graph replay of the bs-1 evaluation forward (sc2bench_amd/graphs.py): bit-identical to the eager forward, the analyzers see
the same object, a parameter update re-captures, other batch sizes / modes keep the eager path."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def bench_mod():
    import bench
    return bench


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def test_graphed_forward_is_the_eager_forward(S, dev, bench_mod):
    model = bench_mod.build_model(dev)
    x = bench_mod.synthetic_batch(6, dev, seed=4)
    with torch.no_grad():
        S.hip.configure(eval_graphs=False)
        model.analyzes_after_compress = True
        model.analyzers = [S.FileSizeAnalyzer(unit='KB')]
        model.activate_analysis()
        eager = [model(x[i:i + 1]).clone() for i in range(6)]
        sizes_eager = list(model.analyzers[0].file_size_list)
        enc_eager = [model.bottleneck_layer.encode(x[i:i + 1])['strings'][0][0] for i in range(6)]
        model.clear_analysis()
        S.hip.configure(eval_graphs=True)
        try:
            graphed = [model(x[i:i + 1]).clone() for i in range(6)]
            assert model.__dict__.get('_eval_graphs_error') is None, model.__dict__.get('_eval_graphs_error')
            g = model._eval_graphs_for(x[0:1])
            assert g is not None, 'the bs-1 forward did not take the graphs'
            sizes_graphed = list(model.analyzers[0].file_size_list)
            assert sizes_graphed == sizes_eager                       # the analyzers saw the same pickled objects
            for a, b in zip(eager, graphed):
                assert torch.equal(a, b)
            # replay order does not matter, outputs are clones (a later replay does not overwrite an earlier result)
            again = model(x[2:3])
            assert torch.equal(again, eager[2]) and torch.equal(graphed[2], eager[2])
            # the symbols graph A leaves are the eager encoder's: the same byte string
            sym_h = g.symbols(x[3:4])
            strings, _ = S.hip.rans_encode_host(model.bottleneck_layer.entropy_bottleneck._host_tables(), sym_h,
                                                index_div=g.latent_shape[0] * g.latent_shape[1])
            assert strings[0] == enc_eager[3]
            # batches beyond eval_graph_max_batch run eagerly; the result is the concatenation either way
            assert model._eval_graphs_for(x[0:2]) is None
            assert torch.equal(model(x[0:2]), torch.cat(eager[0:2]))
            # a parameter update invalidates the capture: the next forward re-captures and follows the new weights
            with torch.no_grad():
                model.fc.bias.add_(1.0)
            moved = model(x[0:1])
            assert torch.allclose(moved.float(), eager[0].float() + 1.0, atol=2e-2) and not torch.equal(moved, eager[0])
            S.hip.configure(eval_graphs=False)
            assert torch.equal(model(x[0:1]), moved)
        finally:
            S.hip.configure(eval_graphs=True)
            model.deactivate_analysis()


def test_training_mode_and_f32_encoder_keep_the_eager_path(S, dev, bench_mod):
    model = bench_mod.build_model(dev)
    x = bench_mod.synthetic_batch(1, dev, seed=1)
    with torch.no_grad():
        model(x)
        assert model._eval_graphs_for(x) is not None
        model.set_encoder_precision('f32')
        assert model._eval_graphs_for(x) is None and '_eval_graphs' not in model.__dict__
        model.set_encoder_precision('bf16')
        assert model._eval_graphs_for(x) is not None
        model.train()
        assert '_eval_graphs' not in model.__dict__
        model.eval()
    # under autograd the eager path runs (a graph replay records nothing)
    assert model._eval_graphs_for(x) is None
