"""The timed region of bench.py: K batches through the package's stage pipeline (sc2bench_amd/pipeline.py), bracketed as the
contract says, and the backend-side rank count."""
import time

import torch
import torch.distributed as dist


WORKLOAD_PIPELINE = {   # (coder group G, coder streams) per workload, by measurement
    # (3 coder streams, not 4: tools/attic/r06_pipeline_sweep6.sh, 7 alternating runs each -- K = 20 median 52.1 k against 51.1 k images/s, and
    #  the 4-stream runs dropped to 49 k at K = 100 on one box where the 3-stream runs held 52.5 k)
    'es224': (8, 3),
    # mean-scale hyperprior: the per-symbol-index decoder holds 120 KB of LDS per workgroup for ~20 ms; 2 048 streams per launch with
    # two 16-stream waves per workgroup (64 CUs held, the library's choice from 1 024 streams up), three launches in flight
    # (tools/attic/mshp_sweep.sh, 40 steps: 35.2 k images/s at G = 2, 36.6 k at G = 8 with one wave per workgroup, 39.3 k with two)
    'mshp224': (8, 3),
    'fp_input': (8, 4),      # 32 streams per batch: 256 per launch
    'seg513': (8, 4),        # 16 streams x 393 k symbols per batch: 128 per launch, ~85 ms of chain each way
    'det800x1216': (8, 6),   # 6 streams x 1.45 M symbols per batch: 48 per launch (one wave), ~330 ms each way
}


def make_pipeline(args, model, dev):
    import sc2bench_amd as S
    pipe = _make_pipeline(args, model, dev, S)
    pipe.host_ramp_skip = bool(getattr(args, 'host_ramp_skip', 1))
    return pipe


def _make_pipeline(args, model, dev, S):
    g_default, c_default = WORKLOAD_PIPELINE[args.workload]
    return S.StagePipeline(model, dev, coder_group=args.coder_group or g_default, coder_streams=args.inflight or c_default,
                           max_inflight=args.max_inflight, ramp=bool(args.ramp), lag=max(0, args.lag),
                           front_priority=args.front_priority, back_priority=args.back_priority, coder_priority=args.coder_priority,
                           back_streams=max(1, args.split_mfma), share_buffer=not args.cat_symbols,
                           coder_kwargs={'dequantized': False} if args.unfused_dequantize else None,
                           # (auto only behind a warm-up: the host path's pinned staging -- 450 MB of cudaHostAlloc -- must not be
                           #  allocated inside a timed region that starts cold)
                           host_steps=(None if getattr(args, 'warmup', 1) > 0 else 0) if getattr(args, 'host_steps', -1) < 0 else args.host_steps)


def timed_pipeline_run(pipe, x, steps, select, distributed, timeline=False):
    """K batches through the package's stage pipeline (sc2bench_amd/pipeline.py), bracketed as the contract says: the caller has
    synchronised; this starts the clock, issues K batches, synchronises every stream (+ barrier) and stops it.
    -> (elapsed s, host issue s, KernelTimer, last (output, nbytes, status), record)"""
    from sc2bench_amd import hip
    rec = {'timeline': []} if timeline else {}
    last = [None]

    def keep(step, out, nb, st):
        last[0] = (out, nb, st)

    with hip.KernelTimer(select) as timer:
        t0 = time.perf_counter()
        pipe.run(x, n_steps=steps, on_output=keep, record=rec)
        t_issued = time.perf_counter()
        pipe.synchronize()
        if distributed:
            dist.barrier()
        t1 = time.perf_counter()
    return t1 - t0, t_issued - t0, timer, last[0], rec


def ranks_reduced(dev, distributed):
    """RCCL-side proof of the rank count: every rank contributes 1 to a device all-reduce on the backend (the process group's
    world size in `config.process_group` comes from the launcher's environment)."""
    if not distributed:
        return None
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(one.item()))
