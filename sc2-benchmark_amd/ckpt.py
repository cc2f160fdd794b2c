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


def load_ckpt(ckpt_file_path, model=None, optimizer=None, lr_scheduler=None, strict=True):
    if ckpt_file_path is None or not os.path.isfile(ckpt_file_path):
        logger.info('ckpt file path is None or does not exist: {}'.format(ckpt_file_path))
        return None, None
    ckpt = torch.load(ckpt_file_path, map_location='cpu', weights_only=False)
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
