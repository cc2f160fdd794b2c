"""The `bs1_eval` row of the bench line: the reference's evaluation mode (batch size 1, host bytes)."""
import time

import torch


def bs1_eval(model, x, dev, n=64):
    """The reference's evaluation mode (script/task/image_classification.py:106-145, test batch size 1): per image
    forward() = encode -> FileSizeAnalyzer on the pickled {'strings','shape'} -> decode -> head, through the host API
    (bytes objects cross to the host and back, as in the reference).  Round 6: the two device halves of that forward replay HIP
    graphs (sc2bench_amd/graphs.py) around the host range coder; the row also carries the eager figure, the launch count of an
    eager forward and per-image latency percentiles (one synchronize per image)."""
    import sc2bench_amd as S
    from sc2bench_amd import hip
    model.analyzes_after_compress = True
    model.analyzers = [S.FileSizeAnalyzer(unit='KB')]
    model.activate_analysis()

    def loop(count):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(count):
            model(x[i % x.shape[0]:i % x.shape[0] + 1])
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    with torch.no_grad():
        graphs_on = bool(hip.host_policy.eval_graphs)
        hip.configure(eval_graphs=False)
        for i in range(3):
            model(x[i:i + 1])
        with hip.KernelTimer() as kt:           # every tagged launch of ONE eager forward
            model(x[0:1])
            torch.cuda.synchronize(dev)
        launches = len(kt.records)
        dt_eager = loop(n)
        hip.configure(eval_graphs=graphs_on)
        for i in range(3):
            model(x[i:i + 1])
        used = model.__dict__.get('_eval_graphs') is not None and any(isinstance(v, S.graphs.EvalGraphs) for v in model.__dict__['_eval_graphs'].values())
        model.clear_analysis()
        dt = loop(n)
        sizes = model.analyzers[0].file_size_list[-n:]
        lat = []
        for i in range(n):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            model(x[i % x.shape[0]:i % x.shape[0] + 1])
            torch.cuda.synchronize(dev)
            lat.append(1e3 * (time.perf_counter() - t0))
        lat.sort()
    model.deactivate_analysis()
    return {'images_per_s': n / dt, 'ms_per_image': 1e3 * dt / n, 'images': n,
            'data_size_kb_mean': sum(sizes) / len(sizes),
            'latency_ms': {'p50': lat[len(lat) // 2], 'p99': lat[min(len(lat) - 1, int(0.99 * len(lat)))], 'what': 'one synchronize per image'},
            'hip_graphs': ('2 graph replays per image (encoder | dequantise + decoder + layer2..fc) around the host range coder, '
                           'captured once per input shape in this process' if used else
                           'off: ' + str(model.__dict__.get('_eval_graphs_error') or 'policy')),
            'eager': {'ms_per_image': 1e3 * dt_eager / n, 'launches_per_image': launches},
            'what': 'bs 1, forward() with host bytes (encode -> pickle size -> decode -> layer2..fc), one stream'}
