"""Checkpoint tooling on the host: the reference-format checkpoint written by the imported reference
(tests/golden/make_golden.py tooling) is read with the safe loader and its architecture recovered from shapes."""
import os

import numpy as np
import torch

import nerf_fixtures as fx
from nerf_amd import checkpoint as ck

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(GOLD, "ref_ckpt_small.tar")


def test_reference_checkpoint_architecture():
    c = ck.read_checkpoint(CKPT)
    assert c["global_step"] == 2
    for key in ("network_fn_state_dict", "network_fine_state_dict"):
        a = ck.infer_nerf_arch(c[key])
        assert a == dict(D=3, W=128, input_ch=63, input_ch_views=27, output_ch=4, skips=[1], use_viewdirs=True)
    assert len(c["optimizer_state_dict"]["state"]) == 28
    text = ck.describe(CKPT)
    assert "global_step 2" in text and "76100 parameters" in text


def test_infer_default_architectures():
    sd = {k: torch.from_numpy(v) for k, v in fx.make_nerf_state(0).items()}
    assert ck.infer_nerf_arch(sd) == dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4],
                                          use_viewdirs=True)
    sd = {k: torch.from_numpy(v) for k, v in fx.make_nerf_state(0, input_ch_views=0, output_ch=5,
                                                                use_viewdirs=False).items()}
    a = ck.infer_nerf_arch(sd)
    assert (a["use_viewdirs"], a["output_ch"], a["input_ch_views"]) == (False, 5, 0)
    sd = {k: torch.from_numpy(v) for k, v in fx.make_dnerf_state(0).items()}
    assert ck.is_dnerf(sd)
    a = ck.infer_dnerf_arch(sd)
    assert (a["D"], a["W"], a["D_deform"], a["W_deform"]) == (8, 256, 8, 256)
    assert (a["input_ch"], a["input_ch_views"], a["input_ch_time"]) == (63, 27, 21)
    assert a["skips"] == [4] and a["skips_deform"] == [4] and a["use_viewdirs"] and not a["use_layernorm"]


def test_multires_from_channels():
    assert ck._multires(63) == 10 and ck._multires(27) == 4 and ck._multires(21, dims=1) == 10
    try:
        ck._multires(64)
    except ValueError:
        pass
    else:
        raise AssertionError("64 channels is not a default embedding width")


def test_save_is_the_reference_layout(tmp_path):
    from nerf_amd.network import NeRF
    m = NeRF(D=3, W=128, input_ch=63, input_ch_views=27, output_ch=5, skips=[1], use_viewdirs=True)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    p = str(tmp_path / "000002.tar")
    ck.save_checkpoint(p, 2, m, None, opt)
    ref = ck.read_checkpoint(CKPT)
    mine = ck.read_checkpoint(p)
    assert list(mine["network_fn_state_dict"]) == list(ref["network_fn_state_dict"])
    assert "network_fine_state_dict" not in mine and set(mine) <= set(ref)
    for k, v in ref["network_fn_state_dict"].items():
        assert mine["network_fn_state_dict"][k].shape == v.shape


def test_ray_table_golden_is_float64_rounded_once():
    g = np.load(os.path.join(GOLD, "tooling.npz"))
    # NumPy >= 2 promotes get_rays_np to float64 for train.py's float64 K; the table is cast afterwards (train.py:153)
    assert str(g["table_dtype_before_cast"][0]) == "float64"
    H, W = g["table_hw"]
    assert g["table"].shape == (3 * H * W, 3, 3) and g["table"].dtype == np.float32


def test_bench_finds_its_committed_counter_evidence():
    """bench.py's roofline.traffic / mfma_busy_frac come from the committed PMC summaries: the kernel names it asks for must
    resolve against them (kernels gain template parameters between rounds; the lookup follows the argument-list prefix)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for workload, kernel in (("render", "query_fwd_kernel<8, false, false, false>"),
                             ("render_bf16", "query_fwd_bf16_kernel<8, true>"),
                             ("render_bf16x3", "query_fwd_bf16x3_kernel<8, true>")):
        t = bench.pmc_traffic(workload, kernel)
        assert t is not None and 1e8 < t < 1e9, (workload, t)          # ~115 MB algorithmic per 4 M-point launch
        u = bench.pmc_mfma_util(workload, kernel)
        assert 0.5 < u["mfma_busy_frac"] < 1.0 and 1.5 < u["clock_ghz"] < 2.5, (workload, u)
    table = {"k<1, true, false>": 1, "k<1, true, true>": 2, "j<2>": 3}
    assert bench._kernel_entry(table, "k<1, true>") == 1 and bench._kernel_entry(table, "j<2>") == 3


def test_straightforward_checkpoint_is_recognised(tmp_path):
    """A DNeRF_Straightforward checkpoint has NeRF's keys; checkpoint.networks_from_checkpoint tells it apart by the saved
    args (vanilla_dynamic/model.py:189-199) or, without them, by layer 0's fan-in (63 + 21 is not 3 + 6 L)."""
    import torch
    from nerf_amd import checkpoint
    import nerf_amd.vanilla_dynamic as vd
    m = vd.DNeRF_Straightforward(D=4, W=64, input_ch=63, input_ch_views=27, input_ch_time=21, skips=[1], use_viewdirs=True)
    sd = m.state_dict()
    arch = checkpoint.infer_straight_arch(sd)
    assert arch == dict(D=4, W=64, input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=4, skips=[1], use_viewdirs=True)
    assert checkpoint.infer_straight_arch(sd, dict(network_type="straightforward", multires=10, multires_time=10)) == arch
    assert checkpoint.infer_straight_arch(sd, dict(network_type="deformation")) is None
    from nerf_amd.network import NeRF
    assert checkpoint.infer_straight_arch(NeRF(4, 64, 63, 27, 4, [1], True).state_dict()) is None
    for args in (None, dict(network_type="straightforward", multires=10, multires_time=10, i_embed=0)):
        path = tmp_path / f"s{0 if args is None else 1}.tar"
        blob = {"global_step": 7, "network_fn_state_dict": sd, "network_fine_state_dict": sd}
        if args:
            blob["args"] = args
        torch.save(blob, path)
        got = checkpoint.networks_from_checkpoint(str(path), device="cpu")
        assert isinstance(got["network_fn"], vd.DNeRF_Straightforward) and isinstance(got["network_fine"], vd.DNeRF_Straightforward)
        assert got["network_fn"].input_ch_time == 21 and got["global_step"] == 7
        assert all(torch.equal(a, b) for a, b in zip(got["network_fn"].state_dict().values(), sd.values()))
