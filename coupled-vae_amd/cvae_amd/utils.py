"""The three utils.utils functions on the hot path (CV/utils/utils.py:10-21,69-82)."""
import torch

from . import ops


def gpu_wrapper(item):
    """CV/utils/utils.py:10-15 (config.gpu = torch.cuda.is_available())."""
    return item.cuda() if torch.cuda.is_available() else item


def sort_resort(length):
    """CV/utils/utils.py:18-21: stable descending order + inverse, as Python lists (the
    reference's return type).  Computed on the device by cvae_sort_resort."""
    s_idx, res_idx = ops.sort_resort(length.to(torch.int64))
    return s_idx.tolist(), res_idx.tolist()


def log_sum_exp(value, dim=None, keepdim=False):
    """CV/utils/utils.py:69-82.  Compatibility helper for callers outside the fused IW path
    (inside it the log-sum-exp is part of cvae_iw_reduce / cvae_mi)."""
    if dim is not None:
        m, _ = torch.max(value, dim=dim, keepdim=True)
        value0 = value - m
        if keepdim is False:
            m = m.squeeze(dim)
        return m + torch.log(torch.sum(torch.exp(value0), dim=dim, keepdim=keepdim))
    m = torch.max(value)
    return m + torch.log(torch.sum(torch.exp(value - m)))
