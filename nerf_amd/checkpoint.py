"""Checkpoint tooling for the reference's on-disk format.

Static NeRF (train.py:260-265):            {'global_step', 'network_fn_state_dict', 'network_fine_state_dict',
                                            'optimizer_state_dict'}
D-NeRF (vanilla_dynamic/model.py:189-211): the same plus 'args' (vars(args)); 'network_fine_state_dict' optional.

Files are read with torch.load(weights_only=True) only: tensors, numbers, strings and containers -- nothing that
executes code from the file.  The architecture is recovered from the tensor shapes, so a checkpoint can go straight
to the packed kernel layouts (`packed_from_checkpoint`) without the training script's argument parser.

    python -m nerf_amd.checkpoint <file.tar>      # prints what the file holds
"""
import sys

import torch


def read_checkpoint(path, map_location="cpu"):
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(ckpt, dict) or "network_fn_state_dict" not in ckpt:
        raise ValueError(f"{path}: not a NeRF checkpoint (no 'network_fn_state_dict')")
    return ckpt


def _mlp_arch(sd, prefix, n_extra=0):
    """(D, W, input_ch, skips) of a `pts_linears`-style stack `prefix.{i}.weight` whose layer after a skip takes
    W + input_ch inputs (network.py:29-32); n_extra = input channels that are not re-fed at skips (D-NeRF time)."""
    D = 0
    while f"{prefix}.{D}.weight" in sd:
        D += 1
    if D == 0:
        raise ValueError(f"no '{prefix}.0.weight' in the state dict")
    W, in0 = (int(v) for v in sd[f"{prefix}.0.weight"].shape)
    skips = [i - 1 for i in range(1, D) if int(sd[f"{prefix}.{i}.weight"].shape[1]) != W]
    for s in skips:
        if int(sd[f"{prefix}.{s + 1}.weight"].shape[1]) != W + in0 - n_extra:
            raise ValueError(f"{prefix}.{s + 1}.weight: unexpected fan-in for a skip layer")
    return D, W, in0, skips


def infer_nerf_arch(sd):
    """Constructor arguments of network.NeRF (network.py:8-44) from a state dict."""
    D, W, input_ch, skips = _mlp_arch(sd, "pts_linears")
    use_viewdirs = "feature_linear.weight" in sd      # views_linears exists either way (network.py:33)
    input_ch_views = int(sd["views_linears.0.weight"].shape[1]) - W
    # the viewdirs head ignores output_ch (network.py:36-42), so create_nerf's 5 is not recoverable: 4 is equivalent
    output_ch = 4 if use_viewdirs else int(sd["output_linear.weight"].shape[0])
    return dict(D=D, W=W, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=output_ch, skips=skips,
                use_viewdirs=use_viewdirs)


def infer_dnerf_arch(sd):
    """Constructor arguments of vanilla_dynamic.network.DNeRF_Deformation from a state dict (keys `_occ.*`, `_time.*`,
    `_time_out.*`; vanilla_dynamic/network.py:205-252)."""
    occ = {k[len("_occ."):]: v for k, v in sd.items() if k.startswith("_occ.")}
    a = infer_nerf_arch(occ)
    input_ch_time = int(sd["_time.0.weight"].shape[1]) - a["input_ch"]
    # a skip layer of the time net re-reads the position block only (vanilla_dynamic/network.py:296-299)
    D_t, W_t, _, skips_t = _mlp_arch(sd, "_time", n_extra=input_ch_time)
    return dict(D=a["D"], W=a["W"], D_deform=D_t, W_deform=W_t, input_ch=a["input_ch"],
                input_ch_views=a["input_ch_views"], input_ch_time=input_ch_time, output_ch=a["output_ch"],
                skips=a["skips"], skips_deform=skips_t, use_viewdirs=a["use_viewdirs"],
                use_layernorm=any("norm" in k.lower() for k in sd))


def is_dnerf(sd):
    return any(k.startswith("_occ.") for k in sd)


def infer_straight_arch(sd, args=None):
    """Constructor arguments of vanilla_dynamic.network.DNeRF_Straightforward (network.py:100-149) from a state dict, or
    None when the file is a static NeRF.  Its keys are NeRF's; what differs is layer 0's fan-in, cat[gamma(x), gamma(t)]:
    told by the saved `args` (network_type, multires, multires_time) when the checkpoint has them (vanilla_dynamic/model.py:
    189-199 saves vars(args)), otherwise by a fan-in that is not 3 + 6 L but (3 + 6 L) + (1 + 2 Lt), Lt = 10 preferred."""
    if "pts_linears.0.weight" not in sd:
        return None
    a = infer_nerf_arch(sd)
    total = a["input_ch"]
    args = args or {}
    tc = None
    if args.get("network_type") == "straightforward":
        Lt = int(args.get("multires_time", 10))
        tc = 1 + 2 * Lt if Lt > 0 else 1
        if not args.get("no_include_input", False) and "multires" in args and int(args.get("i_embed", 0)) == 0:
            assert total == 3 + 6 * int(args["multires"]) + tc, "checkpoint args and layer-0 fan-in disagree"
    elif args.get("network_type") is None and (total - 3) % 6 != 0:
        for Lt in [10] + list(range(9, -1, -1)):
            if total - (1 + 2 * Lt) >= 3 and (total - (1 + 2 * Lt) - 3) % 6 == 0:
                tc = 1 + 2 * Lt
                break
    if tc is None:
        return None
    return dict(D=a["D"], W=a["W"], input_ch=total - tc, input_ch_views=a["input_ch_views"], input_ch_time=tc,
                output_ch=a["output_ch"], skips=a["skips"], use_viewdirs=a["use_viewdirs"])


def _multires(ch, dims=3):
    """L of get_embedder from a channel count dims + 2*L*dims (positional_encoding.py:39-64, include_input=True)."""
    L, rem = divmod(ch - dims, 2 * dims)
    if ch and (rem or L < 0):
        raise ValueError(f"{ch} channels are not {dims} + {2 * dims}*L: not the default embedder")
    return max(L, 0)


def networks_from_checkpoint(path, device="cuda", precision="fp32"):
    """-> dict(global_step, network_fn, network_fine, optimizer_state_dict, args): modules of this package with the
    file's weights; network_fine is None when the file has none."""
    ckpt = read_checkpoint(path)

    def build(sd):
        straight = None if is_dnerf(sd) else infer_straight_arch(sd, ckpt.get("args"))
        if straight is not None:
            from .vanilla_dynamic.network import DNeRF_Straightforward
            m = DNeRF_Straightforward(**straight)
        elif is_dnerf(sd):
            from .vanilla_dynamic.network import DNeRF_Deformation
            from .vanilla_dynamic.positional_encoding import get_embedder
            arch = infer_dnerf_arch(sd)
            m = DNeRF_Deformation(**arch)
            m.embed_fn = get_embedder(_multires(arch["input_ch"]), 0, input_dims=3)[0]
        else:
            from .network import NeRF
            m = NeRF(**infer_nerf_arch(sd))
        m.load_state_dict(sd, strict=True)
        m.precision = precision
        return m.to(device)

    fine = ckpt.get("network_fine_state_dict")
    return dict(global_step=int(ckpt.get("global_step", 0)), network_fn=build(ckpt["network_fn_state_dict"]),
                network_fine=build(fine) if fine else None,
                optimizer_state_dict=ckpt.get("optimizer_state_dict"), args=ckpt.get("args"))


def packed_from_checkpoint(path, which="network_fn", device="cuda", precision="fp32"):
    """A static-NeRF checkpoint straight to the packed kernel layout -> ops.PackedNet / PackedNetBf16 / PackedNetBf16x3 (query-ready,
    raw points in, positional encoding fused).  which: 'network_fn' | 'network_fine'."""
    from . import ops
    sd = read_checkpoint(path)[f"{which}_state_dict"]
    if is_dnerf(sd):
        raise ValueError("D-NeRF checkpoints hold two chained nets: use networks_from_checkpoint")
    a = infer_nerf_arch(sd)
    params = {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in sd.items()}
    L, Lv = _multires(a["input_ch"]), _multires(a["input_ch_views"]) if a["use_viewdirs"] else 0
    known = {"fp32": ops.PackedNet, "bf16": ops.PackedNetBf16, "bf16x3": ops.PackedNetBf16x3}
    if precision not in known:
        raise ValueError(f"precision must be one of {sorted(known)}, not {precision!r}")
    cls = known[precision]
    kw = dict(multires=L, multires_views=Lv)
    if precision == "fp32":
        kw["mode"] = 0
    net = cls(params, a["D"], a["W"], a["input_ch"], a["input_ch_views"], a["output_ch"], a["skips"], a["use_viewdirs"],
              **kw)
    net.refresh()
    return net


def save_checkpoint(path, global_step, network_fn, network_fine, optimizer, args=None):
    """train.py:260-265 (static; args None) / vanilla_dynamic/model.py:189-211 (D-NeRF; args given)."""
    d = {"global_step": global_step, "network_fn_state_dict": network_fn.state_dict(),
         "optimizer_state_dict": optimizer.state_dict()}
    if network_fine is not None:
        d["network_fine_state_dict"] = network_fine.state_dict()
    if args is not None:
        d["args"] = vars(args) if not isinstance(args, dict) else args
    torch.save(d, path)


def describe(path):
    ckpt = read_checkpoint(path)
    lines = [f"{path}: global_step {ckpt.get('global_step')}"]
    for key in ("network_fn_state_dict", "network_fine_state_dict"):
        sd = ckpt.get(key)
        if not sd:
            continue
        arch = infer_dnerf_arch(sd) if is_dnerf(sd) else infer_nerf_arch(sd)
        n = sum(v.numel() for v in sd.values())
        lines.append(f"  {key}: {'DNeRF_Deformation' if is_dnerf(sd) else 'NeRF'} {arch} ({n} parameters)")
    opt = ckpt.get("optimizer_state_dict")
    if opt:
        g = opt["param_groups"][0]
        lines.append(f"  optimizer: {len(opt['state'])} tensors with state, lr {g.get('lr')}, betas {g.get('betas')}")
    if "args" in ckpt:
        lines.append(f"  args: {len(ckpt['args'])} entries")
    return "\n".join(lines)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(describe(p))
