"""Checkpoint helpers (reference: effdet/helpers.py, timm.models.load_checkpoint)."""
import logging
from collections import OrderedDict

import torch


def load_pretrained(model, url, filter_fn=None, strict=True):
    """The reference fetches `url` with torch.hub (helpers.py:14-22).  This build runs offline: a local
    file path is loaded, an http(s) URL raises."""
    if not url:
        logging.warning('Pretrained model URL is empty, using random initialization.')
        return
    if url.startswith('http://') or url.startswith('https://'):
        raise RuntimeError('no network: download %s yourself and pass checkpoint_path=' % url)
    state_dict = torch.load(url, map_location='cpu')
    if filter_fn is not None:
        state_dict = filter_fn(state_dict)
    model.load_state_dict(state_dict, strict=strict)


def load_checkpoint(model, checkpoint_path, use_ema=False, strict=True):
    """timm.models.load_checkpoint semantics: accepts a bare state-dict or a dict with
    'state_dict' / 'state_dict_ema'; strips a leading 'module.'."""
    ckpt = torch.load(checkpoint_path, map_location='cpu')
    key = 'state_dict_ema' if use_ema else 'state_dict'
    sd = ckpt[key] if isinstance(ckpt, dict) and key in ckpt else ckpt
    out = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())
    model.load_state_dict(out, strict=strict)
