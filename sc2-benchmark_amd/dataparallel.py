"""Data parallelism for the bottleneck path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

The path shards by image -- every image is an independent forward and an independent rANS stream -- so inference /
evaluation needs no data-path collective; training adds ONE gradient all-reduce per step over the
trainable set.  In stage 1 of the Entropic-Student recipe only `bottleneck_layer` trains (reference config
`splitable_resnet50-fp-beta0.08_from_resnet50.yaml:135`): 1 304 168 f32 values = 5.2 MB, latency-bound on a ring
(2*(7/8)*5.2 MB / 153 GB/s ~ 60 us of wire time), hence a single flat bucket and a single RCCL call instead of
DDP's per-bucket hooks; larger trainable sets (stage 2, ~106 MB) are split into `bucket_mb` buckets, each launched
from a post-accumulate-grad hook when its last gradient lands, so they overlap what remains of backward
(`FlatGradAllReducer.overlap`).

Replaces `torch.nn.parallel.DistributedDataParallel` as the reference wraps it
(script/task/image_classification.py:110-111; configs `wrapper: 'DistributedDataParallel'`).
"""
import os

import torch
import torch.distributed as dist


def collectives_active(process_group=None):
    """True when this module's collectives should actually be issued: a process group exists and either has more than one
    rank or `SC2_DP_WORLD1_COLLECTIVES=1` asks for them on a single rank too.  The second form exists for ONE purpose: a box with
    one GPU can then run every collective of the training / evaluation path on RCCL (world 1: `init_process_group('nccl')`,
    broadcast, the hook-launched bucket all-reduces, the metric reductions on HIP tensors) -- tests/test_00_rccl_gpu.py."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(process_group) > 1 or os.environ.get('SC2_DP_WORLD1_COLLECTIVES') == '1'


def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK (torchrun) -> (distributed, rank, world, device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_cuda = torch.cuda.is_available()
    device = torch.device('cuda', local_rank) if use_cuda else torch.device('cpu')
    if use_cuda:
        torch.cuda.set_device(local_rank)
    launched = all(k in os.environ for k in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'))   # torchrun, any world size
    if (world > 1 or (launched and os.environ.get('SC2_DP_WORLD1_COLLECTIVES') == '1')) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if use_cuda else 'gloo')
        kwargs = {'device_id': device} if (use_cuda and backend == 'nccl') else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return dist.is_initialized() and collectives_active(), rank, world, device


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _visible_index(local_rank):
    visible = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES')
    if visible:
        ids = [v.strip() for v in visible.split(',') if v.strip()]
        if local_rank < len(ids) and ids[local_rank].isdigit():
            return int(ids[local_rank])
    return local_rank


def _numa_of_pci(sysfs_root, bdf):
    with open(os.path.join(sysfs_root, 'bus', 'pci', 'devices', bdf, 'numa_node')) as f:
        node = int(f.read().strip())
    return node if node >= 0 else None


def gpu_numa_node(local_rank, sysfs_root='/sys', dev_root='/dev'):
    """NUMA node of the `local_rank`-th GPU, read from sysfs -- NO HIP call (a rank binds itself before it touches the device).
    Two sources, the first that answers: (i) the KFD topology: GPUs are its nodes with SIMDs, in node order = the runtime's device
    order, `location_id` + `domain` = the PCI address (readable by root only on some kernels: on this pool's boxes it is not);
    (ii) the render nodes this process can open (`/dev/dri/renderD*`: in a container only the GPUs it was given), each resolved
    through `/sys/class/drm/<node>/device` to its PCI address, AMD devices only, in PCI order.  HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES re-map the index.  -> int >= 0, or None (unknown, one node, not a bare-metal topology)."""
    index = _visible_index(local_rank)
    try:
        nodes_dir = os.path.join(sysfs_root, 'class', 'kfd', 'kfd', 'topology', 'nodes')
        gpus = []
        for name in sorted(os.listdir(nodes_dir), key=int):
            props = {}
            with open(os.path.join(nodes_dir, name, 'properties')) as f:
                for ln in f:
                    k, _, v = ln.strip().partition(' ')
                    props[k] = v
            if int(props.get('simd_count', '0')) > 0:
                gpus.append(props)
        if 0 <= index < len(gpus):
            loc, dom = int(gpus[index]['location_id']), int(gpus[index].get('domain', '0'))
            return _numa_of_pci(sysfs_root, '{:04x}:{:02x}:{:02x}.{:x}'.format(dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    except (OSError, ValueError, KeyError):
        pass
    try:
        bdfs = []
        for name in os.listdir(os.path.join(dev_root, 'dri')):
            if not name.startswith('renderD'):
                continue
            link = os.path.join(sysfs_root, 'class', 'drm', name, 'device')
            bdf = os.path.basename(os.path.realpath(link))
            with open(os.path.join(sysfs_root, 'bus', 'pci', 'devices', bdf, 'vendor')) as f:
                if f.read().strip().lower() != '0x1002':
                    continue
            bdfs.append(bdf)
        bdfs.sort()
        if 0 <= index < len(bdfs):
            return _numa_of_pci(sysfs_root, bdfs[index])
    except (OSError, ValueError):
        pass
    return None


def bind_rank_to_gpu_numa(local_rank, sysfs_root='/sys', dev_root='/dev'):
    """Pins the calling process (every thread it starts later inherits the mask: the host coder's pool, the data-loader workers, the
    launch thread) to the CPUs of the NUMA node its GPU hangs off.  Eight ranks on a two-socket node otherwise share whatever
    cores the scheduler picks, and a rank whose launch thread sits on the far socket pays the inter-socket hop on every
    doorbell and every pinned-buffer copy.  Call BEFORE the first HIP call.  -> {'numa_node', 'cpus'} or None (nothing bound:
    unknown topology, a single node, or an affinity mask the launcher already narrowed to other CPUs -- that one is respected)."""
    import os
    node = gpu_numa_node(local_rank, sysfs_root, dev_root)
    if node is None or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        with open(os.path.join(sysfs_root, 'devices', 'system', 'node', 'node{}'.format(node), 'cpulist')) as f:
            cpus = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return {'numa_node': node, 'cpus': len(allowed)}
    except (OSError, ValueError):
        return None


def shard_range(n_items, rank, world):
    """Contiguous [start, end) slice of n_items for this rank (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class FlatGradAllReducer(object):
    """Keeps the gradients of the trainable parameters in flat f32 buckets (each p.grad is a view) and averages them
    across ranks with one all-reduce per bucket.

    Buckets are filled in REVERSE parameter order (the order in which backward produces gradients).  Inside
    ``with reducer.overlap(): loss.backward()`` a post-accumulate-grad hook counts the gradients that have landed in
    each bucket and launches the bucket's all-reduce (async, on the collective's own stream) the moment its last
    gradient is written, so the first buckets travel while backward is still producing the rest (stage 2 of the
    recipe: ~106 MB).  ``all_reduce()`` launches whatever was not launched by a hook -- every bucket when overlap()
    was not used, and buckets holding a parameter that received no gradient in this backward (e.g. `quantiles`, whose
    gradient comes from the separate aux-loss backward) -- then waits for all of them and divides by the world size.
    Gradients accumulated by earlier backward passes of the same step (aux loss) are part of what is reduced."""

    def __init__(self, params, bucket_mb=25.0, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = collectives_active(process_group)   # (world > 1, or a forced single-rank group: collectives_active)
        cap = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self.buckets = []
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flats = []
        self._bucket_of = {}
        for b, bucket in enumerate(self.buckets):
            dev, n = bucket[0].device, sum(p.numel() for p in bucket)
            flat = torch.zeros(n, dtype=torch.float32, device=dev)
            o = 0
            for p in bucket:
                p.grad = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
                self._bucket_of[p] = b
            self.flats.append(flat)
        self._armed = False
        self._pending = [0] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self.launched_by_hook = 0      # buckets whose all-reduce started inside backward (last step)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] \
            if self.active else []

    def _launch(self, b):
        self._works[b] = dist.all_reduce(self.flats[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        if not self._armed:
            return
        b = self._bucket_of[p]
        self._pending[b] -= 1
        if self._pending[b] == 0 and self._works[b] is None:
            self._launch(b)
            self.launched_by_hook += 1

    def overlap(self):
        """Context manager around the LAST backward of a step: buckets launch as they fill."""
        reducer = self

        class _Armed(object):
            def __enter__(self_inner):
                reducer._pending = [len(b) for b in reducer.buckets]
                reducer._works = [None] * len(reducer.buckets)
                reducer.launched_by_hook = 0
                reducer._armed = reducer.active
                return reducer

            def __exit__(self_inner, *exc):
                reducer._armed = False
                return False
        return _Armed()

    def zero_grad(self):
        for flat in self.flats:
            flat.zero_()

    def all_reduce(self):
        """Average gradients over ranks.  Call after backward(); a no-op on a single process."""
        if not self.active:
            return
        for b in range(len(self.buckets)):
            if self._works[b] is None:
                self._launch(b)
        for b, flat in enumerate(self.flats):
            self._works[b].wait()
            flat.div_(self.world)
        self._works = [None] * len(self.buckets)

    def nbytes(self):
        return sum(f.numel() * 4 for f in self.flats)


def broadcast_parameters(module, src=0, process_group=None):
    """Makes every rank start from rank `src`'s parameters and buffers (what DDP does at construction)."""
    if not collectives_active(process_group):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.numel() > 0:
            dist.broadcast(t.data, src=src, group=process_group)


def collective_device(process_group=None):
    """The device a collective's tensors must live on for the backend of this process group: the current HIP device under
    RCCL ("nccl" cannot reduce host tensors), the host under gloo."""
    if dist.is_initialized() and 'nccl' in str(dist.get_backend(process_group)).lower():
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_reduce_mean_scalars(values, device=None, process_group=None):
    """Mean over ranks of per-rank scalars (every rank weighs the same).  `device` None = the backend's device."""
    t = torch.tensor(values, dtype=torch.float64, device=collective_device(process_group) if device is None else device)
    if collectives_active(process_group):
        dist.all_reduce(t, group=process_group)
        t /= dist.get_world_size(process_group)
    return t.tolist()


def all_reduce_sum_scalars(values, device=None, process_group=None):
    """Sum over ranks, on the backend's device.  Metric reduction as the reference's MetricLogger does it
    (`synchronize_between_processes`: every meter's [count, total] are summed, the global average is total / count
    afterwards), so ranks that saw different numbers of samples weigh by their samples."""
    t = torch.tensor(values, dtype=torch.float64, device=collective_device(process_group) if device is None else device)
    if collectives_active(process_group):
        dist.all_reduce(t, group=process_group)
    return t.tolist()


def all_gather_picklable(data, process_group=None):
    """Every rank's picklable object, in rank order (script/task/coco/eval.py:161-200, the C3 collective of SURVEY.md 2.3:
    variable-size pickled buffers, sizes exchanged first, payloads padded to the longest).  `torch.distributed`'s object
    collective does exactly that on the backend's device (RCCL: the current HIP device; gloo: host)."""
    if not collectives_active(process_group):
        return [data]
    out = [None] * dist.get_world_size(process_group)
    dist.all_gather_object(out, data, group=process_group)
    return out


def merge_coco_eval(img_ids, eval_imgs, process_group=None):
    """Per-rank (image ids, evalImgs array [K, A, I_rank]) -> the merged, id-sorted, de-duplicated pair every rank
    needs before COCOeval.accumulate (script/task/coco/eval.py:203-223)."""
    import numpy as np
    all_ids = all_gather_picklable(list(img_ids), process_group)
    all_eval = all_gather_picklable(eval_imgs, process_group)
    merged_ids = np.array([i for part in all_ids for i in part])
    merged_eval = np.concatenate(list(all_eval), 2)
    merged_ids, idx = np.unique(merged_ids, return_index=True)
    return merged_ids, merged_eval[..., idx]
