"""Helpers of the reference's vec2wav/utils.py that sit on the generator path (utils.py:23-26, 35-36)
plus the generator checkpoint format (utils.py:39-58, train.py:227-230)."""
import glob
import os

import torch


def get_padding(kernel_size, dilation=1):
    """`int((k*d - d)/2)` - vec2wav/utils.py:35-36."""
    return int((kernel_size * dilation - dilation) / 2)


def init_weights(m, mean=0.0, std=0.01):
    """vec2wav/utils.py:23-26.  On a weight-normed conv this touches nothing that survives the next
    forward (the reference applies it AFTER weight_norm); kept for surface parity."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1 and hasattr(m, 'weight') and isinstance(getattr(m, 'weight'), torch.Tensor):
        m.weight.data.normal_(mean, std)


def load_checkpoint(filepath, device):
    """utils.py:39-44: torch.load of a `g_%08d` file -> {'generator': state_dict}."""
    assert os.path.isfile(filepath)
    return torch.load(filepath, map_location=device)


def save_checkpoint(filepath, obj):
    """utils.py:47-50."""
    torch.save(obj, filepath)


def scan_checkpoint(cp_dir, prefix):
    """utils.py:53-58: lexicographically last `prefix????????` file, or None."""
    cp_list = glob.glob(os.path.join(cp_dir, prefix + '????????'))
    if len(cp_list) == 0:
        return None
    return sorted(cp_list)[-1]
