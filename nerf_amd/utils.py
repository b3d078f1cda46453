"""Small helpers the reference keeps in utils.py:9-16 and that its drivers import by name: the default device, the
DEBUG switch of render_rays' NaN scan, the two loss / metric formulas and the 8-bit conversion for written images."""
import math

import numpy as np
import torch

# 'cuda' is the HIP device on a ROCm build of PyTorch; there is no CPU path behind these modules, the fallback only
# keeps imports working on a machine without a GPU (CPU tests of the host logic)
device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
DEBUG = False          # True: render_rays scans its outputs for NaN / Inf after every chunk (costs host syncs)
_LN10 = math.log(10.0)


def img2mse(x, y):
    """Mean squared error over every element, as train.py:232-239 uses it for the rgb and rgb0 losses."""
    diff = x - y
    return torch.mean(diff ** 2)


def mse2psnr(x):
    """PSNR in dB of an MSE tensor for images in [0, 1]: -10 log10(x).  (The reference divides by a one-element tensor
    holding ln 10; a Python float gives the same values without a device constant.)"""
    return -10.0 * torch.log(x) / _LN10


def to8b(x):
    """[0, 1] floats (numpy) -> uint8 by clipping and truncation, the conversion behind every PNG the drivers write."""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)
