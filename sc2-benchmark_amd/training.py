"""Training driver for the Entropic-Student recipe, driven by the reference's own YAML `train:` blocks.

The reference hands this to torchdistill's `DistillationBox` (script/task/image_classification.py:153-193; behaviour
restated in SURVEY.md appendix C).  What the hot path needs from it is small and is reproduced here with the same
config keys: per-model `sequential` / `frozen_modules` / `forward_hook` / `requires_grad`, the forward-hook IO
dict, `WeightedSumLoss` over `SimpleLossWrapper(MSELoss | ...)` and mid-level losses (`BppLoss`), optimizer /
scheduler by key, the separate `aux_loss` backward (image_classification.py:74-77), and -- instead of DDP -- the
flat-bucket gradient all-reduce of `dataparallel.py`.
"""
from collections import OrderedDict

import torch
from torch import nn

from . import hip
from .dataparallel import FlatGradAllReducer
from .loss import MIDDLE_LEVEL_LOSS_DICT


def hip_epi_bias_relu():
    return hip.EPI_BIAS_RELU


def hip_nhwc_input(x, cin_pad):
    """NCHW batch (any float dtype / layout) -> bf16 NHWC with the channels zero-padded to `cin_pad`."""
    if x.shape[1] == cin_pad and x.dtype == torch.bfloat16:
        return x.permute(0, 2, 3, 1).contiguous()
    return hip.nchw_f32_to_nhwc_bf16(x.float().contiguous(), cin_pad)


def get_module(root, path):
    """'a.b.0' -> submodule ('.' = the model itself)."""
    if path in ('.', ''):
        return root
    mod = root
    for part in path.split('.'):
        mod = mod[int(part)] if isinstance(mod, (nn.Sequential, nn.ModuleList)) and part.isdigit() \
            else getattr(mod, part)
    return mod


def redesign_model(model, sequential):
    """`sequential: [names]` -> nn.Sequential over the named children (torchdistill redesign_model); empty -> model."""
    if not sequential:
        return model
    return nn.Sequential(OrderedDict((name.replace('.', '__'), get_module(model, name)) for name in sequential))


def freeze(model, frozen_paths):
    for path in frozen_paths or list():
        for p in get_module(model, path).parameters():
            p.requires_grad_(False)


class ForwardHookManager(object):
    """io_dict[module_path] = {'input': ..., 'output': ...}, refilled on every forward, cleared after the loss."""

    def __init__(self, model, hook_config):
        self.io_dict = dict()
        self.handles = []
        paths_in = (hook_config or {}).get('input') or []
        paths_out = (hook_config or {}).get('output') or []
        self._paths_out = set(paths_out) - set(paths_in)     # modules whose OUTPUT alone is wanted
        self.all_paths = set(paths_out) | set(paths_in)
        for path in sorted(set(paths_in) | set(paths_out)):
            module = get_module(model, path)
            self.handles.append(module.register_forward_hook(self._make(path, path in paths_in, path in paths_out)))

    def _make(self, path, want_in, want_out):
        def hook(module, inputs, output):
            entry = self.io_dict.setdefault(path, dict())
            if want_in:
                entry['input'] = inputs[0] if len(inputs) == 1 else inputs
            if want_out:
                entry['output'] = output
        return hook

    def io_dict_paths(self):
        """paths a caller may fill in place of the module hook (output-only hooks)."""
        return self._paths_out

    def pop(self):
        out, self.io_dict = self.io_dict, dict()
        return out

    def clear(self):
        for h in self.handles:
            h.remove()
        self.handles = []


class KDLoss(nn.Module):
    """alpha * CE(student, labels) + (1 - alpha) * T^2 * KL(student / T || teacher / T)  (torchdistill KDLoss)."""

    def __init__(self, student_module_path='.', student_module_io='output', teacher_module_path='.',
                 teacher_module_io='output', temperature=1.0, alpha=0.5, reduction='batchmean', **kwargs):
        super().__init__()
        self.sp, self.sio, self.tp, self.tio = student_module_path, student_module_io, teacher_module_path, \
            teacher_module_io
        self.temperature, self.alpha, self.reduction = temperature, alpha, reduction

    def forward(self, student_io_dict, teacher_io_dict, targets=None, *args, **kwargs):
        s = student_io_dict[self.sp][self.sio].float()
        t = teacher_io_dict[self.tp][self.tio].float()
        T = self.temperature
        soft = nn.functional.kl_div(torch.log_softmax(s / T, dim=1), torch.softmax(t / T, dim=1),
                                    reduction=self.reduction)
        if self.alpha is None or self.alpha == 0 or targets is None:
            return soft
        hard = nn.functional.cross_entropy(s, targets, reduction='mean' if self.reduction == 'batchmean' else self.reduction)
        return self.alpha * hard + (1 - self.alpha) * (T ** 2) * soft


class SimpleLossWrapper(nn.Module):
    """low-level criterion(input, target) with both taken from the io dicts."""

    def __init__(self, low_level_loss, input, target, **kwargs):
        super().__init__()
        self.low_level_loss = low_level_loss
        self.input_cfg, self.target_cfg = input, target

    @staticmethod
    def _extract(cfg, student_io_dict, teacher_io_dict):
        io = teacher_io_dict if cfg['is_from_teacher'] else student_io_dict
        return io[cfg['module_path']][cfg['io']]

    def forward(self, student_io_dict, teacher_io_dict, targets=None, *args, **kwargs):
        x = self._extract(self.input_cfg, student_io_dict, teacher_io_dict)
        y = self._extract(self.target_cfg, student_io_dict, teacher_io_dict)
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.bfloat16:
            from .frozen import mse_fast_path      # two bf16 feature maps: one HIP pass instead of two casts + loss + grad
            fast = mse_fast_path(self.low_level_loss, x, y)
            if fast is not None:
                return fast
        return self.low_level_loss(x.float(), y.float())


LOW_LEVEL_LOSSES = {'MSELoss': nn.MSELoss, 'L1Loss': nn.L1Loss, 'CrossEntropyLoss': nn.CrossEntropyLoss}


class WeightedSumLoss(nn.Module):
    """sum_i weight_i * term_i(student_io_dict, teacher_io_dict, targets)."""

    def __init__(self, sub_terms=None, model_term=None, **kwargs):
        super().__init__()
        self.terms = nn.ModuleDict()
        self.weights = dict()
        for name, cfg in (sub_terms or {}).items():
            crit_cfg = cfg['criterion']
            key, ckw = crit_cfg['key'], dict(crit_cfg.get('kwargs') or {})
            if 'criterion_wrapper' in cfg:
                wkw = dict(cfg['criterion_wrapper'].get('kwargs') or {})
                term = SimpleLossWrapper(LOW_LEVEL_LOSSES[key](**ckw), **wkw)
            elif key in MIDDLE_LEVEL_LOSS_DICT:
                term = MIDDLE_LEVEL_LOSS_DICT[key](**ckw)
            elif key == 'KDLoss':
                term = KDLoss(**ckw)
            else:
                raise KeyError('criterion `{}` is not available'.format(key))
            self.terms[name] = term
            self.weights[name] = float(cfg.get('weight', 1.0))

    def forward(self, student_io_dict, teacher_io_dict, targets=None):
        total = 0
        for name, term in self.terms.items():
            total = total + self.weights[name] * term(student_io_dict, teacher_io_dict, targets)
        return total


def build_criterion(cfg):
    if cfg['key'] != 'WeightedSumLoss':
        raise KeyError('criterion `{}` is not available'.format(cfg['key']))
    return WeightedSumLoss(**(cfg.get('kwargs') or {}))


class DistillationStage(object):
    """One `train.stageN` block: teacher (frozen, no grad) + student with hooks, criterion, optimizer.

    forward_process(batch, targets) -> loss ; post_forward_process(loss) -> backward, gradient all-reduce, step.
    """

    def __init__(self, teacher, student, stage_config, device, lr_factor=1, head_dtype=None, bucket_mb=25.0):
        self.device = device
        t_cfg, s_cfg = stage_config.get('teacher') or {}, stage_config.get('student') or {}
        self.student_full = student
        self.aux_module = student.get_aux_module() if hasattr(student, 'get_aux_module') else None
        freeze(student, s_cfg.get('frozen_modules'))
        for p in teacher.parameters():
            p.requires_grad_(False)
        teacher.eval()
        self.head_dtype = head_dtype
        self.teacher = redesign_model(teacher, t_cfg.get('sequential'))
        self.student = redesign_model(student, s_cfg.get('sequential'))
        self.autocast_student = False
        if head_dtype is not None:     # bf16 channels_last task-head modules (teacher and the frozen student tail)
            teacher.to(dtype=head_dtype, memory_format=torch.channels_last)
            stepped = {id(p) for p in self.student.parameters() if p.requires_grad}     # what this stage's optimizer updates
            for name in ('layer2', 'layer3', 'layer4', 'avgpool', 'fc'):
                m = getattr(student, name, None)
                if m is None:
                    continue
                if any(id(p) in stepped for p in m.parameters()):
                    # a tail that TRAINS (stage 2) keeps its f32 parameters (optimizer state, weight decay, BatchNorm statistics)
                    # and computes in head_dtype under autocast; only modules this stage does not update are converted
                    self.autocast_student = head_dtype in (torch.bfloat16, torch.float16) and device.type == 'cuda'
                else:
                    m.to(dtype=head_dtype, memory_format=torch.channels_last)
        self.t_hooks = ForwardHookManager(self.teacher, t_cfg.get('forward_hook'))
        self.s_hooks = ForwardHookManager(self.student, s_cfg.get('forward_hook'))
        self.criterion = build_criterion(stage_config['criterion'])
        # as torchdistill does, the optimizer (and the gradient buckets) see the REDESIGNED student: modules left out
        # of `sequential` (avgpool / fc in stage 1) take no part in the step
        params = [p for p in self.student.parameters() if p.requires_grad]
        self.reducer = FlatGradAllReducer(params, bucket_mb=bucket_mb)
        o_cfg = stage_config['optimizer']
        okw = dict(o_cfg.get('kwargs') or {})
        if 'lr' in okw:
            okw['lr'] = okw['lr'] * lr_factor
        self.optimizer = getattr(torch.optim, o_cfg['key'])(params, **okw)
        s = stage_config.get('scheduler')
        self.lr_scheduler = getattr(torch.optim.lr_scheduler, s['key'])(self.optimizer, **(s.get('kwargs') or {})) \
            if s else None
        self.student_full.train()
        for path in s_cfg.get('frozen_modules') or list():   # frozen BatchNorm layers keep their statistics
            get_module(student, path).eval()
        self._frozen_stacks = dict()     # id(module) -> (parameter versions, frozen.FrozenStack)
        self.use_hip_frozen = head_dtype == torch.bfloat16 and device.type == 'cuda'
        if self.use_hip_frozen and hasattr(getattr(student, 'bottleneck_layer', None), 'output_format'):
            # the decoder hands bf16 NHWC features to the frozen tail (and to the layer-1 feature-matching loss) directly
            self._restore_output_format = (student.bottleneck_layer, student.bottleneck_layer.output_format)
            student.bottleneck_layer.output_format = 'bf16_nhwc'

    def _frozen_stack(self, name, module):
        """frozen.FrozenStack of a frozen, eval-mode stack of Bottleneck blocks (None if `module` is not one); rebuilt when a
        parameter or buffer of it changed."""
        from .frozen import FrozenStack
        if not self.use_hip_frozen or not FrozenStack.supported(module):
            return None
        key = tuple(t._version for t in list(module.parameters()) + list(module.buffers()))
        cached = self._frozen_stacks.get(id(module))
        if cached is None or cached[0] != key:
            cached = (key, FrozenStack(name, module))
            self._frozen_stacks[id(module)] = cached
        return cached[1]

    def _teacher_sequence(self):
        """The frozen teacher as a sequence of children whose Bottleneck stacks / stem `_run_sequential` can take: the
        redesigned nn.Sequential itself, or -- for a whole torchvision-layout ResNet (stage 2: `sequential: []`) -- its
        children in forward order with the flatten its forward() does between avgpool and fc."""
        from .resnet import ResNet
        if isinstance(self.teacher, nn.Sequential):
            return self.teacher
        if type(self.teacher) is ResNet and not self.t_hooks.all_paths:
            seq = self.__dict__.get('_teacher_seq')
            if seq is None:
                kids = OrderedDict()
                for name, module in self.teacher.named_children():
                    if name == 'fc':
                        kids['flatten'] = nn.Flatten(1)
                    kids[name] = module
                seq = self.__dict__['_teacher_seq'] = nn.Sequential(kids)
            return seq
        return None

    def _frozen_stem(self, triple, hooks):
        """head._Conv of a (Conv2d without bias, frozen eval-mode BatchNorm, ReLU) run of children that nobody hooks and
        nothing trains, else None."""
        from .head import ConvSpec, _Conv
        from .resnet import FrozenBatchNorm2d
        if not self.use_hip_frozen or len(triple) < 3:
            return None
        (n0, conv), (n1, bn), (n2, act) = triple
        if not (isinstance(conv, nn.Conv2d) and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1) and
                isinstance(bn, (nn.BatchNorm2d, FrozenBatchNorm2d)) and not bn.training and type(act) is nn.ReLU):
            return None
        if any(p.requires_grad for p in list(conv.parameters()) + list(bn.parameters())):
            return None
        if any(q.split('.')[0] in (n0, n1, n2) for q in hooks.all_paths):
            return None
        key = tuple(t._version for t in list(conv.parameters()) + list(bn.parameters()) + list(bn.buffers()))
        cached = self._frozen_stacks.get(id(conv))
        if cached is None or cached[0] != key:
            w = conv.weight.detach().float()
            cin_pad = (w.shape[1] + 7) // 8 * 8
            if cin_pad != w.shape[1]:      # 3 input channels -> 8 (zeros): the kernels read 16-byte channel runs
                w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - w.shape[1], w.shape[2], w.shape[3])], 1)
            cached = (key, _Conv(ConvSpec(w, conv.stride, conv.padding), bn, 'stem'))
            self._frozen_stacks[id(conv)] = cached
        return cached[1]

    def _run_sequential(self, seq, hooks, x, with_grad):
        """`seq(x)` with every frozen Bottleneck stack on the HIP kernels (forward, and input gradient when `with_grad`); the
        forward-hook dict is filled for those stacks as their module hooks would have."""
        from .frozen import FrozenStackFn
        wanted = hooks.io_dict_paths()
        children = list(seq.named_children())
        skip = 0
        for ci, (name, module) in enumerate(children):
            if skip:
                skip -= 1
                continue
            path = name.replace('__', '.')
            is_map = isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4    # (a child may hand on a tuple or a dict)
            stem = self._frozen_stem(children[ci:ci + 3], hooks) if is_map else None
            if stem is not None:      # conv + frozen norm + ReLU at the head of a frozen network (the teacher's stem): one launch
                with torch.no_grad():
                    x_nhwc = hip_nhwc_input(x, stem.w_folded.shape[1])
                    x = stem(x_nhwc, hip_epi_bias_relu()).permute(0, 3, 1, 2)
                skip = 2
                continue
            if (is_map and self.use_hip_frozen and hip.host_policy.maxpool_hip and type(module) is nn.MaxPool2d and x.dtype == torch.bfloat16 and not x.requires_grad and
                    module.dilation in (1, (1, 1)) and not module.ceil_mode and not module.return_indices and x.shape[1] % 8 == 0 and
                    x.is_contiguous(memory_format=torch.channels_last) and not any(q.split('.')[0] == name for q in hooks.all_paths)):
                # the max-pool behind a frozen stem (the teacher's): the library's kernel on the NHWC map (bit-identical to torch's)
                x = hip.maxpool_nhwc(x.permute(0, 2, 3, 1), module.kernel_size, module.stride, module.padding,
                                     tag='maxpool').permute(0, 3, 1, 2)
                continue
            # a hook on the stack's INPUT, or on a module inside it, needs the torch modules to run
            hooked_inside = any(q == path and q not in wanted or q.startswith(path + '.') for q in hooks.all_paths)
            stack = self._frozen_stack(name, module) if (is_map and x.dtype == torch.bfloat16 and not hooked_inside) else None
            if stack is None:
                x = module(x)
                if self.head_dtype is not None and isinstance(x, torch.Tensor) and x.dtype != self.head_dtype and x.dim() == 4:
                    x = x.to(self.head_dtype).contiguous(memory_format=torch.channels_last)
                continue
            if with_grad and x.requires_grad:
                x = FrozenStackFn.apply(x, stack)
            else:
                with torch.no_grad():
                    x = stack.forward(x.permute(0, 2, 3, 1).contiguous())[0].permute(0, 3, 1, 2)
            if path in wanted:
                hooks.io_dict.setdefault(path, dict())['output'] = x
        return x

    def _teacher_forward(self, batch):
        tb = batch.to(self.head_dtype).contiguous(memory_format=torch.channels_last) if self.head_dtype else batch
        with torch.no_grad():
            t_seq = self._teacher_sequence() if self.use_hip_frozen else None
            if t_seq is not None:
                return self._run_sequential(t_seq, self.t_hooks, tb, with_grad=False)
            return self.teacher(tb)

    def forward_process(self, batch, targets=None):
        side = self._teacher_side_stream(batch)
        if side is None:
            t_out = self._teacher_forward(batch)
        else:
            # The frozen teacher shares nothing with the student's forward but the batch: it runs on a stream of its own beside it
            # (its launches are 50 - 100 us kernels of 250 - 1000 workgroups, whose tails leave compute units idle).  The side
            # stream first waits for everything issued so far -- the previous step's backward still reads the previous teacher
            # features, whose blocks this stream's allocator pool is about to hand out again.
            main = torch.cuda.current_stream(batch.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                t_out = self._teacher_forward(batch)
        t_io = self.t_hooks.pop()
        t_io['.'] = {'output': t_out}
        s_out = self._student_forward(batch)
        if side is not None:
            main.wait_stream(side)
        s_io = self.s_hooks.pop()
        s_io['.'] = {'output': s_out}
        return self.criterion(s_io, t_io, targets)

    def _teacher_side_stream(self, batch):
        from . import hip
        if not (self.use_hip_frozen and hip.host_policy.teacher_stream and batch.is_cuda):
            return None
        st = self.__dict__.get('_teacher_stream')
        if st is None:
            st = self.__dict__['_teacher_stream'] = torch.cuda.Stream(device=batch.device)
        return st

    def _student_forward(self, batch):
        if self.autocast_student:
            with torch.autocast(device_type='cuda', dtype=self.head_dtype):
                return self.student(batch)
        if self.head_dtype is None or not isinstance(self.student, nn.Sequential):
            return self.student(batch)
        if self.use_hip_frozen:
            return self._run_sequential(self.student, self.s_hooks, batch, with_grad=True)
        x = batch
        for i, module in enumerate(self.student):   # the bottleneck speaks f32 NCHW; the frozen tail runs in head_dtype
            x = module(x)
            if i == 0:
                x = x.to(self.head_dtype).contiguous(memory_format=torch.channels_last)
        return x

    def post_forward_process(self, loss, bottleneck_updated=False):
        if self.aux_module is not None and not bottleneck_updated:
            self.aux_module.aux_loss().backward()
        with self.reducer.overlap():     # buckets start their all-reduce as backward fills them
            loss.backward()
        self.reducer.all_reduce()
        self.optimizer.step()
        self.reducer.zero_grad()

    def post_epoch_process(self):
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()

    def clean_modules(self):
        self.t_hooks.clear()
        self.s_hooks.clear()
        # the decoder's output format was switched to bf16 NHWC for THIS stage's bf16 tail: a student that goes on with an f32
        # tail must get f32 NCHW features again
        restore = self.__dict__.pop('_restore_output_format', None)
        if restore is not None:
            restore[0].output_format = restore[1]
