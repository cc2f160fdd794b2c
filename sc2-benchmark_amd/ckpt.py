"""Checkpoint I/O with torchdistill's contract: ckpt = {'model', 'optimizer',
'lr_scheduler', 'best_value', 'args'}; a missing path logs and returns (None, None)."""
import logging
import os

import torch

logger = logging.getLogger(__name__)


def save_ckpt(model, optimizer, lr_scheduler, best_value, args, output_file_path):
    d = os.path.dirname(output_file_path)
    if d:
        os.makedirs(d, exist_ok=True)
    ckpt = {'model': model.state_dict(), 'best_value': best_value, 'args': args}
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    if lr_scheduler is not None:
        ckpt['lr_scheduler'] = lr_scheduler.state_dict()
    torch.save(ckpt, output_file_path)


def _torch_load(path, unsafe=False):
    """weights_only load (tensors, containers, argparse.Namespace for the 'args' entry).  A checkpoint that needs
    arbitrary unpickling is refused unless `unsafe` (or SC2_UNSAFE_CKPT=1) says the file is trusted."""
    import argparse
    if unsafe or os.environ.get('SC2_UNSAFE_CKPT') == '1':
        return torch.load(path, map_location='cpu', weights_only=False)
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location='cpu', weights_only=True)
    except Exception as e:
        raise RuntimeError('checkpoint {} does not load with weights_only=True ({}); if the file is trusted, set '
                           'SC2_UNSAFE_CKPT=1 or pass unsafe=True'.format(path, e))


def load_ckpt(ckpt_file_path, model=None, optimizer=None, lr_scheduler=None, strict=True, unsafe=False):
    if isinstance(ckpt_file_path, str) and ckpt_file_path.startswith(('http://', 'https://')):
        # torchdistill downloads here; this build runs offline.  The file is looked up by its basename in
        # $SC2_PRETRAINED_DIR; if it is not there, returning (None, None) would silently leave the model at its random
        # initialisation, so that needs an explicit opt-in (SC2_ALLOW_RANDOM_INIT=1) and still warns.
        root = os.environ.get('SC2_PRETRAINED_DIR')
        local = os.path.join(root, os.path.basename(ckpt_file_path)) if root else None
        if local and os.path.isfile(local):
            ckpt_file_path = local
        elif os.environ.get('SC2_ALLOW_RANDOM_INIT') == '1':
            logger.warning('load_ckpt: {} is a URL and no local copy exists ({}): NOTHING LOADED, the model keeps its '
                           'current (random) parameters'.format(ckpt_file_path, local or 'SC2_PRETRAINED_DIR unset'))
            return None, None
        else:
            raise RuntimeError('load_ckpt: {} is a URL; there is no network here. Put the file at $SC2_PRETRAINED_DIR/{} '
                               'or set SC2_ALLOW_RANDOM_INIT=1 to continue without it'
                               .format(ckpt_file_path, os.path.basename(ckpt_file_path)))
    if ckpt_file_path is None or not os.path.isfile(ckpt_file_path):
        logger.warning('ckpt file path is None or does not exist: {} -- nothing loaded'.format(ckpt_file_path))
        return None, None
    ckpt = _torch_load(ckpt_file_path, unsafe=unsafe)
    if model is not None:
        state = ckpt['model'] if 'model' in ckpt else ckpt
        if strict is None:
            model.load_state_dict(state)
        else:
            try:
                model.load_state_dict(state, strict=strict)
            except TypeError:
                model.load_state_dict(state)
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    if lr_scheduler is not None and 'lr_scheduler' in ckpt:
        lr_scheduler.load_state_dict(ckpt['lr_scheduler'])
    return ckpt.get('best_value', None), ckpt.get('args', None)
