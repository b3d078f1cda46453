"""Counterpart of vanilla_dynamic/utils.py for the symbols the rendering path uses (DEBUG, to8b, device)."""
from ..utils import DEBUG, device, img2mse, mse2psnr, to8b  # noqa: F401
