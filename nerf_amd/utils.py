"""Counterpart of the reference's utils.py:9-16 (device, DEBUG, img2mse, mse2psnr, to8b)."""
import math

import numpy as np
import torch

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
DEBUG = False


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


_LOG10 = math.log(10.0)


def mse2psnr(x):
    return -10.0 * torch.log(x) / _LOG10


def to8b(x):
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)
