"""Evaluation loop of the image-classification task (script/task/image_classification.py:106-145) and the
`-test_only` entry for the BASELINE plumbing configs:

    python -m sc2bench_amd.evaluation --config <yaml> [--device cpu|cuda] [--json '{...}'] [--max_samples 100]

It builds `models.model` (or `models.student_model`) from the reference's YAML file unchanged, loads `test.test_data_loader`
(dataset, sampler, batch size, `collate_fn`), runs the model in eval mode under `torch.inference_mode()`, reports top-1 /
top-5 accuracy and lets every analyzer summarise (data size in KB, as the reference logs it).  Metric totals and counts are
summed over ranks when a process group is up (`MetricLogger.synchronize_between_processes`: SURVEY.md C5).
"""
import argparse
import json
import logging
import time

import torch

from . import config as C
from .analysis import AnalyzableModule
from .dataparallel import all_reduce_sum_scalars

logger = logging.getLogger(__name__)


def compute_accuracy(outputs, targets, topk=(1,)):
    """-> [accuracy@k in percent] (image_classification.py:91-103)."""
    with torch.no_grad():
        maxk = max(topk)
        batch_size = targets.size(0)
        _, preds = outputs.topk(maxk, 1, True, True)
        corrects = preds.t().eq(targets[None])
        return [corrects[:k].flatten().sum(dtype=torch.float32) * (100.0 / batch_size) for k in topk]


class Meter(object):
    def __init__(self):
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.total += float(value) * n
        self.count += n

    @property
    def global_avg(self):
        return self.total / max(1, self.count)


def _pipelined(model, data_loader, device, pipeline):
    """True if evaluate() should run the batches through `pipeline.StagePipeline`: an updated bottleneck model on a HIP device,
    batches of more than one image (the reference measures data size at batch size 1, README.md:100-108: those runs keep the
    per-batch forward, bytes objects and analyzers), and no analyzer that wants the compressed object of every batch."""
    if pipeline is False or device.type != 'cuda':
        return False
    from .pipeline import supports_stages
    if not supports_stages(model):
        return False
    if getattr(model, 'analyzes_after_compress', False) and getattr(model, 'analyzers', None):
        return False
    return pipeline is True or (getattr(data_loader, 'batch_size', None) or 1) > 1


def _evaluate_pipelined(model, data_loader, device, max_samples, pipeline_kwargs):
    """The loop of evaluate() on the stage pipeline: front stages run ahead, the range coder of several batches shares a
    launch on its own HIP stream, top-1 / top-5 hits are counted on the device behind each back stage -- the host reads
    two numbers at the end instead of two per batch.  -> (correct@1, correct@5, samples) as floats."""
    from .pipeline import StagePipeline
    pipe = StagePipeline(model, device, **(pipeline_kwargs or {}))
    # one row of hit counters per back stream: back stage j runs on back stream j % len(back_streams), and two streams adding
    # into the same two addresses would race
    n_back = len(pipe.back_streams)
    hits = torch.zeros(n_back, 2, dtype=torch.float64, device=device)
    targets = {}
    seen = [0]

    def batches():
        for i, (image, target) in enumerate(data_loader):
            if max_samples is not None and seen[0] >= max_samples:
                return
            targets[i] = target.to(device, non_blocking=True)
            seen[0] += len(image)
            yield image.to(device, non_blocking=True)

    def on_output(step, output, nbytes, status):
        # called inside the back stream's context.  The labels were copied on the caller's stream (ordered before this point
        # through the front stage's wait on it -> coder event -> back stream), but the caching allocator only knows the stream a
        # block was ALLOCATED on: without record_stream it would hand the block to a later batch's `target.to(device)` as soon as
        # `target` dies here, while the back stream's eq / topk kernels -- queued milliseconds behind the coder -- still read it
        target = targets.pop(step)
        bs = torch.cuda.current_stream(device)
        target.record_stream(bs)
        row = hits[step % n_back]
        _, preds = output.float().topk(5, 1, True, True)
        corrects = preds.t().eq(target[None])
        row[0] += corrects[:1].sum(dtype=torch.float64)
        row[1] += corrects[:5].sum(dtype=torch.float64)

    record = {}
    pipe.run(batches(), on_output=on_output, record=record)
    pipe.synchronize()
    from .entropy import _raise_on_status
    for st in record['statuses']:       # every coder launch of the run, read once at the end
        _raise_on_status(st, 'evaluate (stage pipeline)')
    h = hits.sum(0).cpu()
    return float(h[0]), float(h[1]), seen[0]


@torch.inference_mode()
def evaluate(model, data_loader, device, max_samples=None, log_freq=1000, title=None, pipeline=None, pipeline_kwargs=None):
    """-> {'acc1', 'acc5', 'samples', 'seconds', 'analysis': [summaries]}.  `pipeline`: None = the stage pipeline
    (sc2bench_amd/pipeline.py) when the model is an updated bottleneck model on a HIP device and the loader's batches hold more
    than one image; True / False force it on / off."""
    model = model.to(device) if device.type == 'cuda' else model
    if hasattr(model, 'use_cpu4compression') and device.type != 'cuda':
        model.use_cpu4compression()
    if title is not None:
        logger.info(title)
    model.eval()
    analyzable = isinstance(model, AnalyzableModule)
    if analyzable:
        model.activate_analysis()
    acc1, acc5 = Meter(), Meter()
    t0 = time.perf_counter()
    seen = 0
    pipelined = _pipelined(model, data_loader, device, pipeline)
    if pipelined:
        c1, c5, seen = _evaluate_pipelined(model, data_loader, device, max_samples, pipeline_kwargs)
        acc1.total, acc1.count = 100.0 * c1, seen       # (Meter: sum of percent x batch size, and the sample count)
        acc5.total, acc5.count = 100.0 * c5, seen
    for i, (image, target) in enumerate(() if pipelined else data_loader):
        if isinstance(image, torch.Tensor):
            image = image.to(device, non_blocking=True)
        if isinstance(target, torch.Tensor):
            target = target.to(device, non_blocking=True)
        output = model(image)
        a1, a5 = compute_accuracy(output.float(), target.to(output.device), topk=(1, 5))
        batch_size = len(image)
        acc1.update(a1.item(), n=batch_size)
        acc5.update(a5.item(), n=batch_size)
        seen += batch_size
        if log_freq and (i + 1) % log_freq == 0:
            logger.info('Test: [{}] acc1 {:.3f} acc5 {:.3f}'.format(i + 1, acc1.global_avg, acc5.global_avg))
        if max_samples is not None and seen >= max_samples:
            break
    # totals and counts are summed over ranks and divided afterwards (MetricLogger.synchronize_between_processes), on the
    # backend's device: an RCCL-only process group cannot reduce a host tensor
    t1, c1, t5, c5, seen_all = all_reduce_sum_scalars([acc1.total, acc1.count, acc5.total, acc5.count, seen])
    top1, top5 = t1 / max(1.0, c1), t5 / max(1.0, c5)
    logger.info(' * Acc@1 {:.4f}\tAcc@5 {:.4f}\n'.format(top1, top5))
    analysis = []
    if analyzable and model.activated_analysis:
        model.summarize()
        analysis = [a.summary() for a in model.analyzers if hasattr(a, 'summary') and getattr(a, 'file_size_list', None)]
    return {'acc1': top1, 'acc5': top5, 'samples': seen, 'samples_all_ranks': int(seen_all), 'seconds': time.perf_counter() - t0, 'analysis': analysis,
            'pipeline': 'stage pipeline (sc2bench_amd/pipeline.py)' if pipelined else 'none: module forward per batch'}


def build_data_loader(dataset_dict, loader_config):
    """`test.test_data_loader` block -> DataLoader (dataset by id, sampler class, kwargs, `collate_fn` by name)."""
    from .transforms import default_collate_w_pil
    dataset = dataset_dict[loader_config['dataset_id']]
    kwargs = dict(loader_config.get('kwargs') or {})
    sampler_cfg = loader_config.get('sampler') or {}
    sampler_cls = sampler_cfg.get('class_or_func')
    sampler = None
    if sampler_cls is not None and not isinstance(sampler_cls, C.Placeholder):
        sampler = sampler_cls(dataset, **(sampler_cfg.get('kwargs') or {}))
    collate = {'default_collate_w_pil': default_collate_w_pil}.get(loader_config.get('collate_fn'))
    kwargs.setdefault('batch_size', 1)
    return torch.utils.data.DataLoader(dataset, sampler=sampler, collate_fn=collate, **kwargs)


def test_only(config, device, max_samples=None, num_workers=None):
    """The `-test_only` branch of main() (image_classification.py:236-250) for a config dict: -> evaluate()'s result."""
    C.import_dependencies(config.get('dependencies'))
    models = config['models']
    model_config = models['student_model'] if 'student_model' in models else models['model']
    model = C.build_model(model_config, device)
    from .ckpt import load_ckpt
    if model_config.get('dst_ckpt') is not None or model_config.get('src_ckpt') is not None:
        load_ckpt(model_config.get('dst_ckpt') or model_config.get('src_ckpt'), model=model, strict=False)
    if hasattr(model, 'update') and not isinstance(model, torch.nn.parallel.DistributedDataParallel):
        from .backbone import check_if_updatable
        if check_if_updatable(model):
            model.update()
    loader_config = dict(config['test']['test_data_loader'])
    if num_workers is not None:
        loader_config['kwargs'] = dict(loader_config.get('kwargs') or {}, num_workers=num_workers)
    loader = build_data_loader(config['datasets'], loader_config)
    return evaluate(model, loader, device, max_samples=max_samples, title='[Student/model]')


def main(argv=None):
    ap = argparse.ArgumentParser(description='test-only evaluation of a reference config on this build')
    ap.add_argument('--config', required=True)
    ap.add_argument('--json', help='json string to overwrite config')
    ap.add_argument('--device', default='cuda' if torch.cuda.is_available() else 'cpu')
    ap.add_argument('--max_samples', type=int)
    ap.add_argument('--num_workers', type=int)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    config = C.load_yaml_file(args.config)
    if args.json:
        C.overwrite_config(config, json.loads(args.json))
    result = test_only(config, torch.device(args.device), args.max_samples, args.num_workers)
    print(json.dumps(result))
    return result


if __name__ == '__main__':
    main()
