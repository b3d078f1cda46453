"""Pieces shared by create_nerf (model.py) and create_dnerf (vanilla_dynamic/model.py): which checkpoint a run resumes
from, how its contents go back into the modules and the fused optimizer, and the train / test keyword sets render()
receives.  Behaviour follows the reference (model.py:55-105, vanilla_dynamic/model.py:118-186), including its console
messages; checkpoints are read with a loader that executes nothing from the file."""
import os

import torch


def checkpoint_candidates(args, prefer_best=False):
    """Paths a run may resume from, oldest first (the last one is used).  --ft_path wins; else `best.tar` alone when
    `prefer_best` (the D-NeRF trainer writes one), else every *tar* file of basedir/expname in name order."""
    explicit = getattr(args, "ft_path", None)
    if explicit is not None and explicit != "None":
        return [explicit]
    folder = os.path.join(args.basedir, args.expname)
    if not os.path.exists(folder):
        print(f"Warning: Checkpoint directory {folder} does not exist.")
        return []
    best = os.path.join(folder, "best.tar")
    if prefer_best and os.path.exists(best):
        print(f"Using best checkpoint: {best}")
        return [best]
    return [os.path.join(folder, name) for name in sorted(os.listdir(folder)) if "tar" in name]


def resume(args, coarse, fine, optimizer, device, prefer_best=False, lenient=False):
    """Loads the newest candidate unless --no_reload; returns the step to continue from (0 for a fresh run).
    lenient (the D-NeRF loader's behaviour): missing keys default, an optimizer state that does not fit is skipped with
    a warning -- enough for rendering."""
    found = checkpoint_candidates(args, prefer_best)
    print("Found ckpts", found)
    if not found or args.no_reload:
        return 0
    path = found[-1]
    print("Reloading from", path)
    state = torch.load(path, map_location=device, weights_only=True)
    if lenient:
        step = state.get("global_step", 0)
        coarse.load_state_dict(state.get("network_fn_state_dict", {}), strict=True)
        if fine is not None and "network_fine_state_dict" in state:
            fine.load_state_dict(state["network_fine_state_dict"])
        try:
            if "optimizer_state_dict" in state:
                optimizer.load_state_dict(state["optimizer_state_dict"])
        except ValueError as err:
            print(f"Warning: Could not load optimizer state (OK for rendering): {err}")
        print(f"[OK] Checkpoint loaded successfully! (step {step})")
        return step
    optimizer.load_state_dict(state["optimizer_state_dict"])
    coarse.load_state_dict(state["network_fn_state_dict"])
    if fine is not None:
        fine.load_state_dict(state["network_fine_state_dict"])
    return state["global_step"]


def render_kwargs(args, query_fn, coarse, fine, **extra):
    """(train, test) keyword sets for render(): the test set is the train set without jitter and density noise."""
    train = dict(network_fn=coarse, network_fine=fine, network_query_fn=query_fn, N_samples=args.N_samples,
                 N_importance=args.N_importance, perturb=args.perturb, raw_noise_std=args.raw_noise_std,
                 white_bkgd=args.white_bkgd, use_viewdirs=args.use_viewdirs)
    train.update(extra)
    test = dict(train, perturb=False, raw_noise_std=0.)
    return train, test


# Arithmetic of the fused kernels, chosen without touching the reference's drivers: an `args.precision` /
# `args.train_precision` attribute if the caller's parser has one (ours to add: the reference's config.py has no such flag),
# else the environment (NRF_PRECISION, NRF_TRAIN_PRECISION).  Unset = the exact f32 kernels.
def apply_precision(args, *modules):
    chosen = {}
    for attr, env in (("precision", "NRF_PRECISION"), ("train_precision", "NRF_TRAIN_PRECISION")):
        v = getattr(args, attr, None) or os.environ.get(env)
        if v:
            chosen[attr] = str(v)
    for m in modules:
        if m is None:
            continue
        for attr, v in chosen.items():
            allowed = getattr(m, f"{attr}_choices", None)
            if allowed is not None and v not in allowed:
                raise ValueError(f"{attr} {v!r} is not one of {sorted(allowed)} for {type(m).__name__}")
            setattr(m, attr, v)
    return chosen
