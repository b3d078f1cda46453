"""A TRAINED field through all three arithmetics, against the imported reference's image of the same weights.

Every other fixture is a random-init or scaled random-init net.  tests/golden/trained_scene.npz (generator:
tests/golden/make_golden.py trained_scene) holds the weights of a 3000-step run of train.py's loop on an analytic scene
(unit sphere with a smooth colour pattern, white background; scripts/train_synthetic_scene.py on the GPU box, train PSNR
7 -> 30 dB), the REFERENCE's float32 and float64 renders of a held-out 200 x 200 view from those weights, and the analytic
image of that view.  This is the closest available stand-in for the north star's "PSNR delta <= 0.05 dB on lego"
(/root/reference/README.md:57): lego data and checkpoints are absent (.MISSING_LARGE_BLOBS), so parity against the README's
dataset numbers is UNPINNED; what is pinned is image-vs-reference-image on identical trained weights, and the PSNR of
each arithmetic against one common target.

On the sphere's silhouette sample_pdf amplifies a 1e-7 difference of a coarse weight into a moved fine sample (the
reference's own float32 and float64 runs agree on 99.90 % of these pixels to 1e-4 and differ by 0.23 on the worst), so the
bars are fractions of the 40 000 pixels plus the PSNR between the images, as VERDICT r2 asked."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    return np.load(os.path.join(ROOT, "tests", "golden", "trained_scene.npz"))


def psnr(a, b):
    return -10.0 * np.log10(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2) + 1e-300)


@pytest.fixture(scope="module")
def images(scene):
    """The held-out view by the fp32, bf16x3 and bf16 kernels from the golden's weights."""
    from nerf_amd.network import NeRF
    from nerf_amd.positional_encoding import get_embedder
    from nerf_amd.render import render, run_network
    nets = []
    for tag in ("coarse", "fine"):
        net = NeRF(8, 256, 63, 27, 5, [4], True)
        net.load_state_dict({k[len(tag) + 1:]: torch.from_numpy(scene[k]) for k in scene.files if k.startswith(tag + ".")},
                            strict=True)
        nets.append(net.cuda())
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    kw = dict(network_query_fn=lambda i, v, f: run_network(i, v, f, e10, e4, 65536), perturb=0., N_importance=128,
              network_fine=nets[1], N_samples=64, network_fn=nets[0], use_viewdirs=True, white_bkgd=True,
              raw_noise_std=0., ndc=False, lindisp=False)
    out = {}
    for prec in ("fp32", "bf16x3", "bf16", "f16"):
        nets[0].precision = nets[1].precision = prec
        with torch.no_grad():
            rgb, disp, acc, _ = render(200, 200, scene["K"], chunk=32768, c2w=torch.from_numpy(scene["c2w"]).cuda(),
                                       near=2., far=6., **kw)
        out[prec] = (rgb.cpu().numpy(), disp.cpu().numpy(), acc.cpu().numpy())
    return out


def test_fp32_image_is_the_references(scene, images):
    rgb, disp, acc = images["fp32"]
    d = np.abs(rgb - scene["rgb"]).max(-1)
    noise = np.abs(scene["rgb"] - scene["rgb_f64"]).max(-1)       # the reference's own f32 rounding on this view
    p = psnr(rgb, scene["rgb"])
    print(f"trained scene fp32 vs reference: within 1e-4 {100 * (d <= 1e-4).mean():.3f} % (reference f32 vs f64: "
          f"{100 * (noise <= 1e-4).mean():.3f} %), worst {d.max():.2e} (reference: {noise.max():.2e}), PSNR between "
          f"the images {p:.1f} dB (reference f32 vs f64: {psnr(scene['rgb'], scene['rgb_f64']):.1f} dB)")
    assert (d <= 1e-4).mean() >= 0.999, (d <= 1e-4).mean()
    assert p >= 80.0, p
    da = np.abs(acc - scene["acc"])
    assert (da <= 1e-4).mean() >= 0.999, (da <= 1e-4).mean()
    # quality of the field itself: the trained net reproduces the analytic view as well as the reference says it does
    assert abs(psnr(rgb, scene["truth"]) - psnr(scene["rgb"], scene["truth"])) <= 0.002


def test_bf16x3_psnr_delta(scene, images):
    """f32-grade arithmetic on the bf16 pipe: |PSNR delta| <= 0.005 dB against the common target, image within 1e-4 of
    the reference's on >= 99.5 % of the pixels."""
    rgb = images["bf16x3"][0]
    p_ref, p_got = psnr(scene["rgb"], scene["truth"]), psnr(rgb, scene["truth"])
    d = np.abs(rgb - scene["rgb"]).max(-1)
    print(f"trained scene bf16x3: PSNR vs analytic {p_got:.4f} dB (reference {p_ref:.4f}), within 1e-4 of the reference "
          f"{100 * (d <= 1e-4).mean():.3f} %, PSNR between the images {psnr(rgb, scene['rgb']):.1f} dB")
    assert 31.0 < p_ref < 34.0, p_ref
    assert abs(p_got - p_ref) <= 0.005, (p_got, p_ref)
    assert (d <= 1e-4).mean() >= 0.995, (d <= 1e-4).mean()


def test_bf16_psnr_delta(scene, images):
    """The north star's bf16 clause: |PSNR delta| <= 0.05 dB, and >= 99.5 % of the pixels within 1e-2 of the reference."""
    rgb = images["bf16"][0]
    p_ref, p_got = psnr(scene["rgb"], scene["truth"]), psnr(rgb, scene["truth"])
    d = np.abs(rgb - scene["rgb"]).max(-1)
    print(f"trained scene bf16: PSNR vs analytic {p_got:.4f} dB (reference {p_ref:.4f}), within 1e-2 of the reference "
          f"{100 * (d <= 1e-2).mean():.3f} %, 99th percentile {np.percentile(d, 99):.2e}, PSNR between the images "
          f"{psnr(rgb, scene['rgb']):.1f} dB")
    assert abs(p_got - p_ref) <= 0.05, (p_got, p_ref)
    assert (d <= 1e-2).mean() >= 0.995, (d <= 1e-2).mean()


def test_f16_psnr_delta(scene, images):
    """The 16-bit kernel with f16 hidden operands: the north star's |PSNR delta| <= 0.05 dB, >= 99.8 % of the pixels within
    1e-2 of the reference, and less than 0.6 x the bf16 kernel's 99th-percentile image error (measured: delta +0.019 dB --
    like bf16's +0.021 it is carried by a handful of silhouette pixels where any rounding moves the resampling --, 99.87 %,
    1.5e-3 vs 3.2e-3, PSNR between the images 58.7 vs 55.3 dB)."""
    rgb, b16 = images["f16"][0], images["bf16"][0]
    p_ref, p_got = psnr(scene["rgb"], scene["truth"]), psnr(rgb, scene["truth"])
    d = np.abs(rgb - scene["rgb"]).max(-1)
    print(f"trained scene f16: PSNR vs analytic {p_got:.4f} dB (reference {p_ref:.4f}), within 1e-2 of the reference "
          f"{100 * (d <= 1e-2).mean():.3f} %, 99th percentile {np.percentile(d, 99):.2e}, PSNR between the images "
          f"{psnr(rgb, scene['rgb']):.1f} dB (bf16: {psnr(b16, scene['rgb']):.1f} dB)")
    assert abs(p_got - p_ref) <= 0.05, (p_got, p_ref)
    assert (d <= 1e-2).mean() >= 0.998, (d <= 1e-2).mean()
    assert np.percentile(d, 99) < 0.6 * np.percentile(np.abs(b16 - scene["rgb"]).max(-1), 99)
    assert psnr(rgb, scene["rgb"]) > psnr(b16, scene["rgb"]) + 2.0
