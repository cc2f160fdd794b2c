"""HIP-graph replay of the updated evaluation forward at small batch sizes.

The reference measures data size -- and evaluates -- at test batch size 1 (script/task/image_classification.py:106-145,
README.md:100-108): `output = model(image)` = encode -> bytes -> analyzers -> decode -> layer2..fc, one image after the other.  On
the GPU that forward is ~45 launches of a few microseconds each around a host range coder (entropy.py: one serial rANS stream
per image is a CPU core's job), and at that size what the device spends is mostly the distance BETWEEN launches: the Python
wrapper and the runtime's launch path per kernel, ~0.3 ms of a 1.95 ms forward (tools/bs1_prof.py).  `EvalGraphs` captures the
two device halves once per input shape --

    A  x  -> encoder (conv0 + GDN, conv2 + GDN, conv4 with the quantiser in its epilogue) -> int32 symbols
    B  int32 symbols -> dequantise -> decoder -> layer2 .. fc -> output

-- and replays them around the same host coder calls: the same kernels with the same arguments in the same order (outputs are
bit-identical to the eager forward, tests/test_gpu_graphs.py), no re-exec, no second process.  The byte strings, the
{'strings', 'shape'} object the analyzers see and the status checks are the eager path's.

A captured graph bakes in device addresses: of the packed weights (re-packed when a parameter's version changes), of the CDF
tables and of its static input / output buffers.  The versions of every parameter and buffer are compared per call (`signature`)
and the model drops its graphs when its storage is re-homed or its mode changes (`invalidate`); the next call captures again.  Persistent kernels take their
work counters from a region of their own when captured (csrc/sc2_common.h: sc2_counter_ring::launch_slot), so a replay never
shares a counter with an eager launch on another stream.
"""
import numpy as np
import torch

from . import hip

__all__ = ['EvalGraphs', 'graphs_for']


def _tensors(model):
    return list(model.parameters()) + list(model.buffers())


def signature(tensors):
    """what a captured graph depends on besides its input: the version of every tensor of the model (25 us for the 304 tensors of
    the ResNet-50 student; walking the module tree for them costs 0.4 ms, so the list is kept with the graphs).  Re-homed storage
    (`.to()`, `load_state_dict`, `update()`, a change of mode or precision) drops the graphs through `invalidate`."""
    return hash(tuple(t._version for t in tensors))


def invalidate(model):
    model.__dict__.pop('_eval_graphs', None)


class EvalGraphs(object):
    """Graphs A and B of ONE input shape of a `SplittableResNet`-like model (attributes: bottleneck_layer with
    analysis / entropy_bottleneck, decode_head)."""

    def __init__(self, model, x):
        bl = model.bottleneck_layer
        eb = bl.entropy_bottleneck
        dev = x.device
        self.shape_key = (tuple(x.shape), x.dtype)
        self.x_static = x.detach().clone()
        medians = eb._median_vector()
        # warm-up on a side stream (weights packed, LDS attributes set, work counters allocated, caching-allocator pools
        # filled): a first launch of a persistent kernel cannot be captured (hipMalloc), and torch asks for it anyway
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                sym = bl.analysis(self.x_static, symbols_for=eb)
                y_hat = hip.eb_dequantize(sym, medians, want_f32=False, want_nhwc=True)[1]
                model.decode_head(y_hat)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.latent_shape = tuple(sym.shape[-2:])
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph_a):
            self.sym_out = bl.analysis(self.x_static, symbols_for=eb)
        self.sym_in = torch.zeros_like(self.sym_out)
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
            y_hat = hip.eb_dequantize(self.sym_in, medians, want_f32=False, want_nhwc=True)[1]
            self.out = model.decode_head(y_hat)
        # pinned staging for the two host transfers (symbols out, decoded symbols in)
        self.sym_host = torch.empty(self.sym_out.shape, dtype=torch.int32, pin_memory=True)
        self.dec_host = torch.empty(self.sym_out.shape, dtype=torch.int32, pin_memory=True)

    def symbols(self, x):
        """graph A on `x`: -> int32 numpy [N, C*h*w] view of the pinned staging buffer (valid until the next call)."""
        self.x_static.copy_(x)
        self.graph_a.replay()
        self.sym_host.copy_(self.sym_out, non_blocking=True)
        torch.cuda.current_stream(x.device).synchronize()
        return self.sym_host.view(self.sym_host.shape[0], -1).numpy()

    def decode_head(self, symbols_host):
        """decoded int32 symbols (numpy [N, C*h*w]) -> the model's output (a clone the caller owns)."""
        np.copyto(self.dec_host.view(self.dec_host.shape[0], -1).numpy(), symbols_host)
        self.sym_in.copy_(self.dec_host, non_blocking=True)
        self.graph_b.replay()
        out = self.out
        return out.clone() if isinstance(out, torch.Tensor) else type(out)((k, v.clone()) for k, v in out.items())


def graphs_for(model, x, max_shapes=4):
    """The EvalGraphs of (model, x's shape), captured on first use; None when graphs do not apply or capture fails (the caller
    runs the eager forward; the reason is kept in `model._eval_graphs_error`)."""
    cache = model.__dict__.setdefault('_eval_graphs', {})
    if 'tensors' not in cache:
        cache['tensors'] = _tensors(model)
    sig = signature(cache['tensors'])
    if cache.get('sig') != sig:
        tensors = cache['tensors']
        cache.clear()
        cache['tensors'], cache['sig'] = tensors, sig
    key = (tuple(x.shape), x.dtype, x.device)
    g = cache.get(key)
    if g is None and key not in cache:
        if len(cache) > max_shapes + 2:      # a loader of ever-changing shapes: stop capturing, run eagerly
            return None
        try:
            g = EvalGraphs(model, x)
            # (capturing allocates packed weights on first use: take the signature again so that the next call matches)
            cache['sig'] = signature(cache['tensors'])
        except Exception as e:      # noqa: BLE001 -- a capture that fails must leave the eager path usable
            model.__dict__['_eval_graphs_error'] = repr(e)
            torch.cuda.synchronize(x.device)
            g = None
        cache[key] = g
    return g
