"""The bare-name shims of dropin/ (what lets the reference's drivers import this implementation unchanged): every
name the reference's train.py / evaluate.py / run_nerf.py / model.py import from these modules resolves to the
nerf_amd object (checked in a fresh interpreter: the module names collide with nothing there)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# module -> names the reference's own files import from it (train.py:12-19, evaluate.py, model.py:5-8, render.py:8-11)
WANTED = {
    "render": ["batchify", "run_network", "batchify_rays", "raw2outputs", "render_rays", "render", "render_path"],
    "network": ["NeRF"],
    "sampling": ["sample_pdf"],
    "positional_encoding": ["get_embedder", "Embedder"],
    "rays": ["get_rays", "get_rays_np", "ndc_rays"],
    "utils": ["img2mse", "mse2psnr", "to8b", "device"],
    "model": ["create_nerf"],
}


def test_dropin_shims_resolve_to_this_implementation():
    code = f"""
import importlib, os, sys
sys.path.insert(0, {os.path.join(ROOT, 'dropin')!r}); sys.path.insert(1, {ROOT!r})
wanted = {WANTED!r}
for mod, names in wanted.items():
    shim = importlib.import_module(mod)
    real = importlib.import_module('nerf_amd.' + mod)
    assert os.path.dirname(shim.__file__).endswith('dropin'), shim.__file__
    for n in names:
        assert getattr(shim, n) is getattr(real, n), (mod, n)
print('ok')
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_utils_formulas():
    import numpy as np
    import torch
    from nerf_amd.utils import img2mse, mse2psnr, to8b
    a, b = torch.tensor([[0.0, 0.5], [1.0, 0.25]]), torch.tensor([[0.5, 0.5], [0.0, 0.25]])
    assert float(img2mse(a, b)) == (0.25 + 1.0) / 4                                   # utils.py:13
    assert abs(float(mse2psnr(torch.tensor(0.01))) - 20.0) < 1e-5                     # utils.py:14: -10 log10
    np.testing.assert_array_equal(to8b(np.array([-0.2, 0.0, 0.5, 1.0, 1.7])), np.array([0, 0, 127, 255, 255], np.uint8))


def test_kernel_shape_reasons():
    """Which architectures the fused kernels take (supported ranges) -- host logic, no GPU: everything else
    runs as a GPU torch composition behind the same module (nerf_amd.network.NeRF.kernel_shape_reason)."""
    from nerf_amd.network import NeRF
    import nerf_amd.vanilla_dynamic as vd
    assert NeRF(8, 256, 63, 27, 4, [4], True).kernel_shape_reason() is None
    assert NeRF(2, 64, 63, 27, 4, [], True).kernel_shape_reason(embedded_rows=True) is None
    assert "W = 300" in NeRF(8, 300, 63, 27, 4, [4], True).kernel_shape_reason()
    assert NeRF(8, 96, 63, 27, 4, [4], True).kernel_shape_reason() is None        # any width up to 256 is fused (whole
    assert NeRF(8, 100, 63, 27, 4, [4], True).kernel_shape_reason() is None       # 32-channel tiles, zero-padded)
    assert "LayerNorm" in vd.NeRF(8, 96, 63, 27, 4, [4], True, use_layernorm=True).kernel_shape_reason()
    assert "D = 17" in NeRF(17, 128, 63, 27, 4, [4], True).kernel_shape_reason()
    assert "output_ch = 9" in NeRF(4, 128, 63, 0, 9, [2], False).kernel_shape_reason()
    assert NeRF(4, 128, 75, 33, 4, [2], True).kernel_shape_reason() is None                  # raw points: the encoder decides
    assert "embedded channels" in NeRF(4, 128, 75, 33, 4, [2], True).kernel_shape_reason(embedded_rows=True)
    m = vd.DNeRF_Deformation(D=8, W=256, D_deform=8, W_deform=96, input_ch=63, input_ch_views=27, input_ch_time=21)
    assert "W_deform = 96" in m.kernel_shape_reason()
    assert vd.DNeRF_Deformation(D=8, W=128, D_deform=8, W_deform=256, input_ch=63, input_ch_views=27,
                                input_ch_time=21).kernel_shape_reason() is None
    s = vd.DNeRF_Straightforward(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, skips=[4])
    assert s._kernel_takes() and s.precision == "fp32"
    assert not vd.DNeRF_Straightforward(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, skips=[2, 5])._kernel_takes()


def test_precision_flags_reach_the_modules(monkeypatch):
    """create_nerf / create_dnerf take the kernels' arithmetic from `args.precision` / `args.train_precision` when the
    caller's parser has them, else from NRF_PRECISION / NRF_TRAIN_PRECISION; a mode a module does not have is refused."""
    import types

    import pytest

    from nerf_amd import _setup
    from nerf_amd.network import NeRF
    from nerf_amd.vanilla_dynamic.network import DNeRF_Deformation
    monkeypatch.delenv("NRF_PRECISION", raising=False)
    monkeypatch.delenv("NRF_TRAIN_PRECISION", raising=False)
    a, b = NeRF(D=2, W=64), NeRF(D=2, W=64)
    assert _setup.apply_precision(types.SimpleNamespace(), a, None, b) == {}
    assert (a.precision, a.train_precision) == ("fp32", "fp32")
    _setup.apply_precision(types.SimpleNamespace(precision="bf16x3", train_precision="bf16x3_w16"), a, None, b)
    assert (a.precision, a.train_precision, b.train_precision) == ("bf16x3", "bf16x3_w16", "bf16x3_w16")
    monkeypatch.setenv("NRF_TRAIN_PRECISION", "bf16x3_a16")
    c = NeRF(D=2, W=64)
    assert _setup.apply_precision(types.SimpleNamespace(train_precision=None), c) == {"train_precision": "bf16x3_a16"}
    assert (c.precision, c.train_precision) == ("fp32", "bf16x3_a16")
    with pytest.raises(ValueError, match="train_precision"):
        _setup.apply_precision(types.SimpleNamespace(), DNeRF_Deformation(D=2, W=64, D_deform=2, W_deform=64))
    with pytest.raises(ValueError, match="precision 'fp64'"):
        _setup.apply_precision(types.SimpleNamespace(precision="fp64"), NeRF(D=2, W=64))
