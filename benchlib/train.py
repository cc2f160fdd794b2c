"""`bench.py --mode train`: one step of the Entropic-Student recipe (stage 1 or stage 2) on synthetic data."""
import json
import time

import torch
import torch.distributed as dist

from .model import shape_workload, synthetic_batch


STAGE1 = {   # train.stage1 of configs/ilsvrc2012/supervised_compression/entropic_student/splitable_resnet50-fp-beta0.08_from_resnet50.yaml
    'teacher': {'sequential': ['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'],
                'forward_hook': {'input': [], 'output': ['layer1', 'layer2', 'layer3', 'layer4']}},
    'student': {'sequential': ['bottleneck_layer', 'layer2', 'layer3', 'layer4'],
                'frozen_modules': ['layer2', 'layer3', 'layer4'],
                'forward_hook': {'input': [], 'output': ['bottleneck_layer', 'layer2', 'layer3', 'layer4',
                                                         'bottleneck_layer.entropy_bottleneck']}},
    'optimizer': {'key': 'Adam', 'kwargs': {'lr': 0.001}},
    'criterion': {'key': 'WeightedSumLoss', 'kwargs': {'sub_terms': dict(
        [('layer{}'.format(i), {'criterion': {'key': 'MSELoss', 'kwargs': {'reduction': 'sum'}},
                                'criterion_wrapper': {'key': 'SimpleLossWrapper', 'kwargs': {
                                    'input': {'is_from_teacher': False,
                                              'module_path': 'bottleneck_layer' if i == 1 else 'layer{}'.format(i), 'io': 'output'},
                                    'target': {'is_from_teacher': True, 'module_path': 'layer{}'.format(i), 'io': 'output'}}},
                                'weight': 1.0}) for i in (1, 2, 3, 4)] +
        [('bpp', {'criterion': {'key': 'BppLoss', 'kwargs': {'entropy_module_path': 'bottleneck_layer.entropy_bottleneck',
                                                             'reduction': 'sum'}}, 'weight': 0.08})])}},
}


STAGE2 = {   # train.stage2 of the same YAML (:231-295): KD loss on the logits, decoder + layer2-4 + fc train, encoder + prior frozen
    'teacher': {'sequential': [], 'frozen_modules': [], 'forward_hook': {'input': [], 'output': []}},
    'student': {'sequential': [], 'frozen_modules': ['bottleneck_layer.encoder', 'bottleneck_layer.entropy_bottleneck'],
                'forward_hook': {'input': [], 'output': []}},
    'optimizer': {'key': 'SGD', 'kwargs': {'lr': 0.001, 'momentum': 0.9, 'weight_decay': 0.0005}},
    'criterion': {'key': 'WeightedSumLoss', 'kwargs': {'sub_terms': {'kd': {'criterion': {'key': 'KDLoss', 'kwargs': {
        'student_module_path': '.', 'student_module_io': 'output', 'teacher_module_path': '.', 'teacher_module_io': 'output',
        'temperature': 1.0, 'alpha': 0.5, 'reduction': 'batchmean'}}, 'weight': 1.0}}}},
}


def train_bench(args, dev, rank, world, distributed, emit=True):
    """Stage-1 Entropic-Student training step: frozen teacher forward, student forward (HIP bottleneck + frozen tail),
    MSE-sum + 0.08 * bits, aux loss, backward on the HIP kernels, ONE flat-bucket gradient all-reduce (RCCL), Adam."""
    import sc2bench_amd as S
    from sc2bench_amd import training as T, dataparallel as dp
    from sc2bench_amd.resnet import resnet50
    torch.manual_seed(0)
    cfg = {'key': 'FPBasedResNetBottleneck', 'kwargs': {'num_bottleneck_channels': 24, 'num_target_channels': 256}}
    student = S.splittable_resnet(cfg, skips_avgpool=False, skips_fc=False).to(dev)
    teacher = resnet50().to(dev)
    if distributed:
        dp.broadcast_parameters(student)
    stage2 = args.stage == 2
    if stage2:      # the reference updates the bottleneck when stage 2 starts (epoch_to_update): round + detach in the student
        shape_workload(student)
        student.update()
    stage = T.DistillationStage(teacher, student, STAGE2 if stage2 else STAGE1, dev, head_dtype=torch.bfloat16)
    x = synthetic_batch(args.bs, dev, seed=rank)
    targets = torch.randint(0, 1000, (args.bs,), generator=torch.Generator().manual_seed(rank)).to(dev) if stage2 else None

    def step():
        loss = stage.forward_process(x, targets)
        stage.post_forward_process(loss, bottleneck_updated=stage2)
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert torch.isfinite(loss)
    # metric reduction as evaluation does it (sum of [count, total] over ranks), on the backend's device
    g_images, g_loss = dp.all_reduce_sum_scalars([float(args.bs * args.steps), float(loss.detach()) * args.bs])
    line = None
    if rank == 0:
        line = ({
            'metric': 'images/s, Entropic-Student ResNet-50 stage-{} training step, 224^2'.format(args.stage), 'value': args.bs * args.steps * world / elapsed,
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': ('stage 2 of the Entropic-Student recipe (KD loss; decoder + layer2-4 + fc train with batch-statistics '
                                    'BatchNorm: {}; encoder + prior frozen, frozen teacher on the HIP stacks)'.format(
                                        'norm layers + ReLU + residual add on bn.hip, the blocks\' convs on the library\'s kernels under autograd'
                                        if (S.hip.host_policy.bn_train_hip and S.hip.host_policy.conv_train_hip) else
                                        'norm layers on bn.hip, convs on torch / MIOpen' if S.hip.host_policy.bn_train_hip else
                                        'on torch / MIOpen ops under bf16 autocast')) if stage2 else
                                   'stage 1 of the Entropic-Student recipe (bottleneck trains, layer2-4 frozen, frozen teacher)',
                       'batch_per_gpu': args.bs, 'global_batch': args.bs * world, 'gradient_all_reduce_bytes': stage.reducer.nbytes(),
                       'sharding': 'images; one flat-bucket all-reduce per step',
                       'process_group': '{} ({} rank{})'.format(dist.get_backend(), world, '' if world == 1 else 's') if distributed else 'none',
                       'collectives_issued': bool(dp.collectives_active()),
                       'gradient_buckets': len(stage.reducer.buckets),
                       'buckets_launched_from_backward_hooks_last_step': stage.reducer.launched_by_hook,
                       'teacher_on_side_stream': bool(S.hip.host_policy.teacher_stream),
                       'gdn_kernels': 'resident-row (gdn512_rows / gdn96_strips)' if S.hip.host_policy.gdn_rows else 'tile GEMMs',
                       'fused_forward_stages': [n for n, on in (('enc.conv0+gdn96', S.hip.host_policy.train_fused_conv0),
                                                                ('enc.conv2+gdn48', S.hip.host_policy.train_fused_conv2),
                                                                ('dec.conv0+igdn512', S.hip.host_policy.train_fused_dec0)) if on]},
            'final_loss': loss.item(), 'images_all_ranks': g_images, 'mean_loss_all_ranks': g_loss / max(g_images / args.steps, 1.0)})
        if emit:
            print(json.dumps(line))
    if distributed and emit:
        dist.barrier()
        dist.destroy_process_group()
    return line
