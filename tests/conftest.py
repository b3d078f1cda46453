import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "static.npz"))


@pytest.fixture(scope="session")
def gold_dnerf():
    return np.load(os.path.join(ROOT, "tests", "golden", "dnerf.npz"))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_exact(a, b, what=""):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = bits(a) != bits(b)
    # NaNs with different payloads still count as equal
    bad &= ~(np.isnan(a) & np.isnan(b))
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} elements differ bitwise; " \
                          f"first at {np.argwhere(bad)[0]}: {a[bad][0]!r} vs {b[bad][0]!r}"


def assert_mostly_close(a, b, atol=1e-4, frac=0.98, hard=2e-4, what="", rtol=0.0, psnr_min=None):
    """End-to-end parity through the hierarchical resampler on the ILL-CONDITIONED ("peaky": sigma head x30, dx x3) test
    nets: sample_pdf's `denom < 1e-5 -> 1` branch (sampling.py:56) is discontinuous, so a 1e-7 difference in a coarse
    weight can move a fine sample by a whole bin on a few rays, and the sharp density turns any 1e-6 shift of a fine
    sample into a visible change.  Require `frac` of the elements within `atol`, all within `hard`, and (images) a PSNR
    floor.  Callers pass frac / hard a small margin from what was MEASURED (printed below, recorded in DESIGN.md
    section 2); well-conditioned nets are held to 1e-4 on every ray instead (tests/test_gpu_wellcond.py)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    fin = np.isfinite(b)
    d = (np.abs(a - b) / (1.0 + (rtol / atol) * np.abs(b)))[fin]     # |a-b| <= atol + rtol |b|
    psnr = -10.0 * np.log10(np.mean((a[fin] - b[fin]) ** 2) + 1e-300)
    print(f"MOSTLY {what}: {(d <= atol).mean():.4f} within {atol:g} (bar {frac}), max {d.max():.3e} (bar {hard:g}), "
          f"PSNR {psnr:.1f} dB" + (f" (bar {psnr_min})" if psnr_min is not None else ""))
    assert (d <= atol).mean() >= frac, f"{what}: only {(d <= atol).mean():.4f} within {atol}; max {d.max():.3e}"
    assert d.max() <= hard, f"{what}: max abs diff {d.max():.3e} > {hard}"
    if psnr_min is not None:
        assert psnr >= psnr_min, f"{what}: PSNR {psnr:.1f} dB < {psnr_min}"


# End-to-end bars on the sharp ("peaky") fixtures, a small margin from what was MEASURED (MI355X and the C oracle agree to
# the digit on the worst rays -- the spread is the reference's conditioning, not this implementation's error): fraction of
# rays within 1e-4, worst ray, PSNR between the outputs.  blender 97.9 % / 1.4e-4 / 93 dB; ndc 98.4 % / 1.8e-4 / 82 dB;
# lindisp (1/z sampling piles the coarse samples up near the camera) 93.8 % / 1.08e-2 / 54.8 dB.
E2E_BARS = {"blender": dict(frac=0.95, hard=5e-4, psnr_min=85.0), "ndc": dict(frac=0.95, hard=5e-4, psnr_min=76.0),
            "lindisp": dict(frac=0.9, hard=2e-2, psnr_min=50.0)}


def assert_samples_close(a, b, atol=1e-4, frac=0.998, what=""):
    """PER-SAMPLE quantities of a fine pass (raw, dx at the 192 merged depths): the resampler's last sample (u = 1
    against a cdf that ends at 1 or at 1 - 1 ulp, sampling.py:29-49) lands in a different bin on a few rays when a coarse
    weight differs by 1e-7, which moves ONE of a ray's samples (usually the deepest ones shift by one slot).  Per-ray
    outputs do not notice and are held to `atol` on every ray by the callers; here `frac` of the samples must agree."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ok = np.abs(a - b) <= atol
    assert ok.mean() >= frac, f"{what}: only {ok.mean():.5f} of the samples within {atol}"
