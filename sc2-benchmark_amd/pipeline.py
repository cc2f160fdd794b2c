"""Software pipeline of the updated bottleneck's evaluation forward over HIP streams.

The reference evaluates batch by batch: `output = model(image)` = encode -> bytes -> decode -> task head, one after the other
(script/task/image_classification.py:106-145 -> sc2bench/models/backbone.py:229-233 -> layer.py:496-521).  On a GPU the range
coder in the middle is a serial state machine per image: ~20 ms of LATENCY for a 224 x 224 latent whatever the batch size,
during which the matrix cores idle.  This module runs the same forward cut into the three stages every updated model of
this package exposes

    stage_front(x[, out=])        encoder (+ hyper transforms) + quantisation        -> (payload, meta)
    stage_coder(payload, meta)    rANS encode to byte streams, then decode them      -> (decoded, nbytes, status)
    stage_back(decoded, meta)     (hyper synthesis +) dequantise + decoder + head    -> output

as an event-driven pipeline: the front stages run ahead on one HIP stream, the coder stages of up to `coder_group`
consecutive batches share ONE launch of the two serial kernels on one of `coder_streams` coder streams (a coder launch is one
wave per 64 image streams: its duration is per-stream latency and does not grow with the stream count; fewer, wider launches
also cost the MFMA kernels beside them less), and each back stage waits for its coder launch on the
decoder + head stream.  Nothing is skipped and nothing is reordered inside a batch: every batch's byte streams are really
produced and really decoded, outputs are bit-identical to the unpipelined forward (tests/test_gpu_pipeline.py).

Round 6, `host_steps`: the FIRST back stage of a run can start no earlier than one coder latency after the first front stage --
~21 ms for a 224 x 224 latent on the device, whatever the batch, during which the run has only front stages to do (a 20-batch run
spent a tenth of its time there).  A model that offers `stage_coder_host` (the FP bottleneck: the same streams coded by the
library's host coder on the CPU cores, ~3 ms for 256 streams on 64 cores plus the two PCIe crossings) gets its first `host_steps`
batches coded there, one batch per group, by one worker thread -- device-to-host copy enqueued at once on a stream of its own, host
encode + decode, host-to-device copy and the dequantising launch on a second stream -- while the device coder takes every later
batch with the usual 1, 2, 4, ... ramp.  Every stream is still really encoded and really decoded, to the same bytes.  Default
(`host_steps=None`): as many batches as the host finishes before the device's first group would (`auto_host_steps`), 0 on a
machine with few cores.  What it took to make that a gain (bs 256, 20 batches, a 2 x 64-core host; profiles/r06l - r06o):
(i) every transfer of those batches is issued by the WORKER thread through the copy engine -- submitted from the thread that issues
the front stages, the second transfer blocked it for 5.7 ms; done by a kernel instead, the copies slowed the front stages beside
them 2 - 3x; (ii) the first early back stage waits, on the device, for the last front stage of the run's opening burst: front stages
that run alone take 0.55 ms, overlapped with a back stage both stretch 2 - 3x, and a run that started its back stages inside the
burst finished no earlier than the all-device run (whose "idle" first 21 ms are where its front stages run alone).  With both:
48.6 -> 49.6 k images/s at K = 20, 51.3 -> 52.3 k at K = 100 (policy `hip.host_policy.pipeline_host_steps`, on).  (iii) Behind h host batches
the back stages are busy for h x 4.3 ms past the burst, so the device coder's first result is not needed one coder latency into the run but
that much later: its ramp starts at 2^h batches per launch instead of 1 (`host_ramp_skip`), i.e. two or three coder launches fewer alive while
the early back stages run: 48.9 - 49.4 -> 50.5 - 51.2 k at K = 20 with three host batches, 51.5 - 51.7 k with four (the default on a host
with >= 64 cores) -- within 2 - 3 % of the 100-batch rate (profiles/r06zz_host_ramp_ab.txt).

`payload` / `decoded` are a tensor or a tuple of tensors whose leading dimension is the batch: the pipeline concatenates
the payloads of a group along it and hands each batch its slice of what the coder returns.  A model whose `stage_front`
accepts `out=` (SplittableResNet on the FP bottleneck: the last encoder conv writes the coder's int32 symbols itself) gets a
row block of the group's shared buffer to write into, so nothing is concatenated.  Batches of a group must agree in `meta`
(the latent's spatial size); a change of shape closes the group early.
"""
import time

import torch

__all__ = ['StagePipeline', 'supports_stages']


def supports_stages(model):
    """True if `model` is an updated bottleneck model that exposes the three stages (and is ready to run them)."""
    return all(callable(getattr(model, n, None)) for n in ('stage_front', 'stage_coder', 'stage_back')) and \
        bool(getattr(model, 'stages_ready', lambda: True)())


def _as_tuple(v):
    return v if isinstance(v, tuple) else (v,)


def _like(v, parts):
    return tuple(parts) if isinstance(v, tuple) else parts[0]


_STREAMS = {}


def pooled_stream(device, role, index, priority=0):
    """The HIP stream a pipeline of this process uses for (role, index, priority) on `device`: created at the first request, handed to
    every StagePipeline that asks later.  The runtime binds streams to GPU_MAX_HW_QUEUES hardware queues in the order they are first USED,
    and two streams on one queue wait for each other; with fresh streams per pipeline, which of a later pipeline's streams shared a queue
    depended on how many streams the process had made before -- the mean-scale-hyperprior row measured behind the headline run read 35.3 k
    images/s with three coder streams in the headline pipeline, 39.5 k with four, 30.2 k without the bs-1 graphs in between, and 39.6 k
    in a process of its own (`tools/attic/r06_mshp_secondary_ab.sh`).  With the pool a later pipeline runs on the queues the first one
    ran on and only adds what it needs beyond them (bench.py measures the bs-1 graphs, which make streams of their own, behind the
    secondary rows for that reason: the detection row's fourth to sixth coder stream read 258 against 333 images/s behind them).
    Binding the whole pool up front gave every row its stand-alone value too, and took 0.7 % off the headline pipeline (nine bound
    streams + the null stream leave no free queue for anything else the process starts).  Pipelines that are alive at the same time
    share these streams, i.e. their stages queue behind each other: still ordered, and not what one process is meant to do."""
    device = torch.device(device)
    dev_i = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev_i, role, int(index), int(priority))
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return st


class StagePipeline(object):
    """Event-driven front / coder-group / back scheduler over any model with `stage_front / stage_coder / stage_back`.

    :param model: updated model in eval mode on `device`
    :param coder_group: G = batches whose streams share one range-coder launch
    :param coder_streams: HIP streams for coder launches (= coder launches that may be in flight)
    :param max_inflight: front stage i waits for back stage i - max_inflight (bounds memory and host run-ahead)
    :param ramp: the first groups of a run hold 1, 2, 4, ... batches, so that the first back stage starts after one coder
                 latency instead of after G front stages
    :param lag: batches between issuing front stage i and back stage i - lag in host order (0: as soon as its coder launch is)
    :param coder_kwargs: keyword arguments of `stage_coder`; None = the model's `stage_coder_kwargs` (FP bottleneck:
                         dequantized=True, the coder's last pass writes the bf16 NHWC latent the decoder reads)
    """

    def __init__(self, model, device, coder_group=8, coder_streams=3, max_inflight=24, ramp=True, lag=0,
                 front_priority=0, back_priority=0, coder_priority=0, back_streams=1, coder_kwargs=None, share_buffer=True,
                 host_steps=None):
        self.model = model
        self.device = torch.device(device)
        self.G = max(1, int(coder_group))
        self.max_inflight = max(1, int(max_inflight))
        self.ramp = bool(ramp)
        self.lag = min(max(0, int(lag)), self.max_inflight - 1)
        self.front_stream = pooled_stream(self.device, 'front', 0, front_priority)
        self.back_streams = [pooled_stream(self.device, 'back', i, back_priority) for i in range(max(1, back_streams))]
        self.coder_streams = [pooled_stream(self.device, 'coder', i, coder_priority) for i in range(max(1, min(int(coder_streams), 13)))]
        self.coder_kwargs = dict(getattr(model, 'stage_coder_kwargs', {}) if coder_kwargs is None else coder_kwargs)
        self.share_buffer = bool(share_buffer) and bool(getattr(model, 'stage_front_takes_out', False))
        self._payload_shapes = {}     # input shape [C, H, W] -> (columns, dtype) of the single-tensor payload (learned per shape)
        # leading batches whose coder stage runs on the host thread pool (None: auto_host_steps at run time; 0: none)
        self.host_steps = host_steps if bool(getattr(model, 'has_stage_coder_host', False)) else 0
        # ONE more stream for those batches: their device-to-host copies (enqueued as the front stages finish), then the
        # host-to-device copies + dequantising launches (enqueued by the worker, later).  One, not two: the runtime maps streams
        # onto GPU_MAX_HW_QUEUES hardware queues, and two streams that share a queue wait for each other -- a back stage once sat
        # 20 ms behind a coder launch of another stream that way (profiles/r06j_timeline_alias.txt)
        self.host_in = self.host_out = pooled_stream(self.device, 'host', 0, 0)
        self._host_staging = {}       # pinned buffers per slot (kept between runs)
        self._worker = None
        self.host_ramp_skip = True

    # ---- plan ------------------------------------------------------------------------------------------------------ #
    @staticmethod
    def auto_host_steps(n_streams, cores=None, device_first_ms=21.0, pcie_ms_per_batch=3.0, ms_per_stream=0.65, limit=4):
        """How many leading batches the host coder should take: batch k leaves the host path at about
        front + copy out + (k + 1) x max(copy, host coding) + copy back; it is worth taking while that is earlier than the device
        coder's first result (one device coder latency).  Host coding of a batch = streams / cores x ~0.65 ms (encode + decode of
        72 600 symbols on one core, csrc/rans_host.cpp); a PCIe crossing of a 256-stream batch ~3 ms.  Few cores (a rank of an
        8-GPU job bound to its NUMA node's share, a small VM): 0."""
        from . import hip
        cores = hip.host_cores() if cores is None else cores
        if cores < 8 or n_streams <= 0:
            return 0
        scale = n_streams / 256.0
        code = n_streams / float(min(cores, 128)) * ms_per_stream
        per_batch = max(code, pcie_ms_per_batch * scale)
        n = 0
        while n < limit and 0.5 + pcie_ms_per_batch * scale * 2 + code + n * per_batch < device_first_ms - 2.0:
            n += 1
        return n

    def group_plan(self, n_steps, host_steps=0):
        """sizes of the coder groups of a run of n_steps batches: `host_steps` single batches for the host coder, then 1, 2, 4, ...
        up to G, then G."""
        sizes = [1] * min(int(host_steps), n_steps)
        g = (1 if self.ramp else self.G)
        if self.ramp and host_steps and self.host_ramp_skip:
            # the host's batches keep the back stages busy for host_steps x ~4.3 ms behind the opening burst: the device coder's
            # first result is not needed after ONE coder latency but after that much more, so its ramp starts wider (fewer
            # coder launches alive while the early back stages run)
            g = min(self.G, 1 << int(host_steps))
        while sum(sizes) < n_steps:
            sizes.append(min(g, self.G, n_steps - sum(sizes)))
            g *= 2
        return sizes

    @staticmethod
    def host_steps_from_measurement(host_ms_per_batch, device_first_ms=21.0, limit=4):
        """The same question as auto_host_steps, answered with what warm() measured on THIS host at this moment (copy out + host coding +
        copy back of one batch, one after the other on the worker): batch k leaves the host path after about (k + 1) x that, and is worth
        taking while this is earlier than the device coder's first result.  A host whose cores are busy with something else, or slower than
        the model assumes, takes fewer batches or none instead of holding the first back stages up."""
        if not host_ms_per_batch or host_ms_per_batch <= 0:
            return limit
        # (warm() times copy out, coding and copy back END TO END; in the run the copies of batch k + 1 ride beside the coding of batch k --
        #  6 - 8 ms measured on the boxes of round 6 where a batch leaves the worker every ~3.5 ms: 0.55 of the measurement)
        return max(0, min(limit, int((device_first_ms - 2.5) // (0.55 * host_ms_per_batch))))

    def resolve_host_steps(self, x):
        if self.host_steps is not None:
            return int(self.host_steps)
        from . import hip
        if not hip.host_policy.pipeline_host_steps:
            return 0
        n = self.auto_host_steps(int(x.shape[0]))
        measured = self.__dict__.get('_host_ms', {}).get(int(x.shape[0]))
        return min(n, self.host_steps_from_measurement(measured)) if measured else n

    def _host_job(self, payload, meta, slot, front_event, timeline, step):
        """worker thread: host coding of one batch; device work on `host_out`.  -> (decoded, nbytes, status, done event)"""
        torch.cuda.set_device(self.device)
        with torch.no_grad(), torch.cuda.stream(self.host_out):
            self.host_out.wait_event(front_event)
            tl0 = None
            if timeline is not None:
                tl0 = torch.cuda.Event(enable_timing=True)
                tl0.record(self.host_out)
            decoded, nb, st = self.model.stage_coder_host(payload, meta, staging=self._host_staging, slot=slot)
            ev = torch.cuda.Event()
            ev.record(self.host_out)
            if tl0 is not None:
                tl1 = torch.cuda.Event(enable_timing=True)
                tl1.record(self.host_out)
                timeline.append(('coder-host', step, tl0, tl1))
        return decoded, nb, st, ev

    def describe(self):
        return {'hip_streams': {'encoder': 1, 'decoder+head': len(self.back_streams), 'range_coder': len(self.coder_streams)},
                'steps_per_coder_launch': self.G, 'max_inflight_steps': self.max_inflight, 'ramp': self.ramp,
                'host_coder_steps': self.__dict__.get('_last_host_steps', self.host_steps)}

    def synchronize(self):
        self.front_stream.synchronize()
        for s in self.back_streams + self.coder_streams + [self.host_out]:
            s.synchronize()
        torch.cuda.synchronize(self.device)

    # ---- the run --------------------------------------------------------------------------------------------------- #
    def run(self, inputs, n_steps=None, on_output=None, record=None):
        """Runs every batch of `inputs` (an iterable of input batches resident on the device, or ONE tensor used for each of
        `n_steps` batches) through the three stages.  `on_output(step, output, nbytes, status)` is called in batch order,
        INSIDE the back stage's stream context, right after that stage has been issued (device work that consumes `output`
        there is ordered behind it; nothing has been waited for).  Returns the number of batches issued; the caller
        synchronises (`synchronize()`).

        `record`: a dict that receives 'statuses' (status vector of every coder launch), 'latency' ([step, front-start event,
        back-end event] of every 8th batch) and, if it holds 'timeline': [], (stage, first step, start, end) event tuples."""
        if isinstance(inputs, torch.Tensor):
            assert n_steps is not None, 'a single input tensor needs n_steps'
            x_single, it = inputs, None
        else:
            x_single, it = None, iter(inputs)
            if n_steps is None and hasattr(inputs, '__len__'):
                n_steps = len(inputs)
        model, dev = self.model, self.device
        timeline = record.get('timeline') if record is not None else None
        statuses = record.setdefault('statuses', []) if record is not None else None
        latency = record.setdefault('latency', []) if record is not None else None

        def tl_event(stream):
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            return e

        host_steps = [None]           # resolved at the first batch (auto: from its stream count)
        plan = [None]
        pending = {}          # step -> (decoded slice, nbytes slice, status slice, meta, coder-done event, whole tensors)
        back_done = {}        # step -> event at the end of its back stage
        group = []            # (step, payload, meta, front-done event, written into the shared buffer?)
        gbuf = [None]
        gshape = [None]       # (rows per batch, columns) the shared buffer was made for
        launches = [0]
        state = {'issued_back': 0}

        def group_target():
            if launches[0] < host_steps[0]:
                return 1
            if plan[0] is not None:
                return plan[0][launches[0]] if launches[0] < len(plan[0]) else self.G
            shift = (launches[0] - host_steps[0]) + (host_steps[0] if self.host_ramp_skip else 0)
            return min(self.G, 1 << min(shift, 16)) if self.ramp else self.G      # (open-ended input: same ramp, no tail trim)

        def flush_host():
            # one batch for the host coder: everything about it -- copy out, host coding, copy back, the dequantising launch --
            # is the worker thread's (a copy-engine transfer submitted while another is in flight can block its caller for
            # milliseconds on this runtime, profiles/r06k_copy_call_stall.txt: that caller must not be the thread that issues the
            # front stages; and copies by a KERNEL slowed the front stages beside them 2 - 3x, profiles/r06n_host_steps_ab.txt)
            slot = launches[0]
            launches[0] += 1
            step, g_pl, meta, g_ev, _ = group[0]
            for t in _as_tuple(g_pl):
                t.record_stream(self.host_out)
            if self._worker is None:
                from concurrent.futures import ThreadPoolExecutor
                self._worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix='sc2-host-coder')
            pending[step] = ('host', self._worker.submit(self._host_job, g_pl, meta, slot, g_ev, timeline, step), meta)
            gbuf[0] = None
            group.clear()

        def flush():
            if launches[0] < host_steps[0] and len(group) == 1:
                return flush_host()
            cs = self.coder_streams[launches[0] % len(self.coder_streams)]
            launches[0] += 1
            with torch.cuda.stream(cs):
                for _, g_pl, _, g_ev, _ in group:
                    cs.wait_event(g_ev)
                    for t in _as_tuple(g_pl):
                        t.record_stream(cs)
                first = group[0][1]
                if len(group) == 1:
                    payload = first
                elif gbuf[0] is not None and all(g[4] for g in group):
                    payload = gbuf[0][:len(group) * group[0][1].shape[0]]     # every front stage wrote its row block: nothing to copy
                    payload.record_stream(cs)
                else:
                    cols = list(zip(*[_as_tuple(g[1]) for g in group]))
                    payload = _like(first, [torch.cat(c) for c in cols])
                gbuf[0] = None
                meta = group[0][2]
                tl0 = tl_event(cs) if timeline is not None else None
                decoded, nb, st = model.stage_coder(payload, meta, **self.coder_kwargs)
                if tl0 is not None:
                    timeline.append(('coder', group[0][0], tl0, tl_event(cs)))
                ev2 = torch.cuda.Event()
                ev2.record(cs)
                if statuses is not None:
                    statuses.append(st)
            sizes = [_as_tuple(g[1])[0].shape[0] for g in group]
            o = 0
            whole = _as_tuple(decoded)
            for (step, _, _, _, _), n in zip(group, sizes):
                sl = slice(o, o + n)
                pending[step] = (_like(decoded, [t[sl] for t in whole]), nb[sl], st[sl], meta, ev2, whole)
                o += n
            group.clear()

        def issue_backs(i, last):
            while state['issued_back'] in pending and (i - state['issued_back'] >= self.lag or last):
                j = state['issued_back']
                if isinstance(pending[j][0], str):
                    # a batch at the host coder: its back stage is issued once the worker has enqueued its result (checked again
                    # after every front stage; the run's end waits for it -- by then there is nothing else to issue)
                    fut = pending[j][1]
                    # ... and, for the first of them, once the run's opening burst of front stages has been issued: that back
                    # stage waits for the burst's last front stage on the device.  Front stages that run ALONE take 0.55 ms each;
                    # overlapped with a back stage and coder launches both stretch 2 - 3x, and a run that starts its back stages
                    # in the middle of the burst finishes no earlier than one that starts them a coder latency later
                    # (profiles/r06l_host_steps_ab.txt).  Behind the burst the early back stages run beside coder launches only.
                    # (a run of known length only: behind a data loader the "burst" is as slow as the loader, and a first result held
                    #  back for 24 batches of JPEG decoding would be a stall, not a saving)
                    burst_last = min(n_steps, self.max_inflight) - 1 if n_steps is not None else 0
                    if not last and (not fut.done() or (j == 0 and i < burst_last)):
                        return
                    dec, nb, st, ev2 = fut.result()
                    if j == 0 and n_steps is not None and state.get('front_ev') is not None:
                        self.back_streams[0].wait_event(state['front_ev'][1])
                    if statuses is not None:
                        statuses.append(st)
                    pending[j] = (dec, nb, st, pending[j][2], ev2, _as_tuple(dec))
                state['issued_back'] += 1
                dec, nb, st, meta, ev2, whole = pending.pop(j)
                bs = self.back_streams[j % len(self.back_streams)]
                with torch.cuda.stream(bs):
                    bs.wait_event(ev2)
                    for t in whole:
                        t.record_stream(bs)
                    nb.record_stream(bs)
                    tl0 = tl_event(bs) if timeline is not None else None
                    out = model.stage_back(dec, meta)
                    if tl0 is not None:
                        timeline.append(('back', j, tl0, tl_event(bs)))
                    if on_output is not None:
                        on_output(j, out, nb, st)
                    back_done[j] = torch.cuda.Event()
                    back_done[j].record(bs)
                    if latency is not None and j % 8 == 0:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record(bs)
                        for r in latency:
                            if r[0] == j:
                                r[2] = e1

        i = 0
        with torch.no_grad():
            x = x_single
            while True:
                if it is not None:
                    try:
                        x = next(it)
                    except StopIteration:
                        break
                    # the batch was produced (copied to the device) on the caller's current stream: the encoder stream waits for
                    # that work and the allocator learns that the encoder stream reads the block
                    self.front_stream.wait_stream(torch.cuda.current_stream(dev))
                    x.record_stream(self.front_stream)
                elif i >= n_steps:
                    break
                if host_steps[0] is None:
                    host_steps[0] = self.resolve_host_steps(x)
                    self.__dict__['_last_host_steps'] = host_steps[0]
                    plan[0] = self.group_plan(n_steps, host_steps[0]) if n_steps is not None else None
                last = (n_steps is not None and i == n_steps - 1)
                if i - self.max_inflight >= state['issued_back']:
                    issue_backs(i, True)      # the run-ahead bound holds for batches at the host coder too: wait for their worker
                with torch.cuda.stream(self.front_stream):
                    if i - self.max_inflight in back_done:
                        # bound the run-ahead of the host and of the encoder stream: memory in flight, latency per batch,
                        # and the caching allocator keeps recycling cross-stream blocks instead of calling hipMalloc
                        back_done.pop(i - self.max_inflight).synchronize()
                    if latency is not None and i % 8 == 0:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record(self.front_stream)
                        latency.append([i, e0, None])
                    tl0 = tl_event(self.front_stream) if timeline is not None else None
                    n = x.shape[0]
                    g_size = group_target()
                    out = None
                    known = self._payload_shapes.get(tuple(x.shape[1:]))     # (columns, dtype) of the payload of such an input
                    if self.share_buffer and g_size > 1 and known is not None:  # (host batches are groups of one: no shared buffer)
                        if gbuf[0] is None and not group:      # one buffer per coder group; front stage k writes row block k
                            gbuf[0] = torch.empty((g_size * n, known[0]), dtype=known[1], device=dev)
                            gshape[0] = (n, known[0])
                        # (a batch of another size or another image shape gets no block: it will close this group and open its own)
                        if gbuf[0] is not None and gshape[0] == (n, known[0]) and gbuf[0].shape[0] >= (len(group) + 1) * n and \
                                all(g[4] for g in group):
                            out = gbuf[0][len(group) * n:(len(group) + 1) * n]
                    if out is not None:
                        payload, meta = model.stage_front(x, out=out)
                    else:
                        payload, meta = model.stage_front(x)
                    if isinstance(payload, torch.Tensor) and payload.dim() == 2:
                        self._payload_shapes[tuple(x.shape[1:])] = (payload.shape[1], payload.dtype)
                    if tl0 is not None:
                        timeline.append(('front', i, tl0, tl_event(self.front_stream)))
                    ev = torch.cuda.Event()
                    ev.record(self.front_stream)
                    state['front_ev'] = (i, ev)
                if group and (group[0][2] != meta or _as_tuple(group[0][1])[0].shape[0] != n):
                    flush()          # another latent shape (or a ragged last batch): this batch opens a new group
                    issue_backs(i, False)
                group.append((i, payload, meta, ev, out is not None))
                if len(group) >= group_target() or last:
                    flush()

                # back stages of every batch whose coder launch has been issued, oldest first: they wait for the coder's
                # event on their own stream, the encoder stream runs ahead
                issue_backs(i, last)
                i += 1
            if group:
                flush()
            issue_backs(i, True)
        assert not pending and not group and state['issued_back'] == i
        return i

    def warm(self, x, n_steps):
        """Resource warm-up, not a step: one range-coder launch per coder group SHAPE of a run of n_steps batches like `x`, on
        the coder stream that group will use, so that the run makes no first-time device allocation (a freshly booted box
        pays ~45 ms for the ~2 GB of workspace / stream buffers of the 8-batch groups otherwise)."""
        with torch.no_grad():
            with torch.cuda.stream(self.front_stream):
                payload, meta = self.model.stage_front(x)
            self.front_stream.synchronize()
            hs = self.resolve_host_steps(x)
            if self.share_buffer and isinstance(payload, torch.Tensor) and payload.dim() == 2:
                # the shared row blocks of the groups that are in flight at once are allocated on the FRONT stream in a run (the caching
                # allocator keeps freed blocks per stream: the coder-stream launches below do not warm them): the first timed region of a
                # process made 2 - 6 hipMalloc calls of 0.6 GB otherwise (torch.cuda.memory_stats, `segment.all.allocated` around the region;
                # none with this -- and no measurable difference in images/s: `tools/attic/r06_warm_blocks_ab.sh`, 5 alternating runs)
                self._payload_shapes[tuple(x.shape[1:])] = (payload.shape[1], payload.dtype)
                sizes = [g for li, g in enumerate(self.group_plan(n_steps, hs)) if li >= hs and g > 1]
                live = sorted(sizes, reverse=True)[:self.max_inflight // max(1, self.G) + 1]
                with torch.cuda.stream(self.front_stream):
                    blocks = [torch.empty((g * payload.shape[0], payload.shape[1]), dtype=payload.dtype, device=payload.device) for g in live]
                    self.model.stage_front(x)      # (... and, while they are held, a front stage's own intermediates: as in the run)
                self.front_stream.synchronize()
                del blocks
            for li, g in enumerate(self.group_plan(n_steps, hs)):
                if li < hs:
                    # the host path's resources: pinned staging of this slot, the worker thread, the host tables
                    with torch.cuda.stream(self.host_out):
                        self.model.stage_coder_host(payload, meta, staging=self._host_staging, slot=li)
                    continue
                cs = self.coder_streams[li % len(self.coder_streams)]
                with torch.cuda.stream(cs):
                    pl = payload if g == 1 else _like(payload, [torch.cat([t] * g) for t in _as_tuple(payload)])
                    self.model.stage_coder(pl, meta, **self.coder_kwargs)
        self.synchronize()
        if hs > 0 and self.host_steps is None:
            # ... and how long THIS host takes for a batch right now, on a slot that has its buffers (host_steps_from_measurement)
            t0 = time.perf_counter()
            with torch.no_grad(), torch.cuda.stream(self.host_out):
                self.model.stage_coder_host(payload, meta, staging=self._host_staging, slot=0)
            self.host_out.synchronize()
            self.__dict__.setdefault('_host_ms', {})[int(x.shape[0])] = 1e3 * (time.perf_counter() - t0)
