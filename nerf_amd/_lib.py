"""ctypes binding of libnrf.so (include/nrf.h).  There is no CPU fallback: if the HIP library is
missing or a call fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnrf.so")
MAX_D = 16
f32p = C.POINTER(C.c_float)
vp = C.c_void_p


class NrfNet(C.Structure):
    """nrf_net_t"""
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("input_ch", C.c_int32), ("input_ch_views", C.c_int32),
                ("output_ch", C.c_int32), ("use_viewdirs", C.c_int32), ("skips_mask", C.c_int32),
                ("reserved", C.c_int32),
                ("pts_w", vp * MAX_D), ("pts_b", vp * MAX_D),
                ("views_w", vp), ("views_b", vp), ("feature_w", vp), ("feature_b", vp),
                ("alpha_w", vp), ("alpha_b", vp), ("rgb_w", vp), ("rgb_b", vp),
                ("out_w", vp), ("out_b", vp),
                ("use_layernorm", C.c_int32), ("reserved2", C.c_int32),
                ("ln_w", vp * MAX_D), ("ln_b", vp * MAX_D), ("views_ln_w", vp), ("views_ln_b", vp)]


class NrfEmbed(C.Structure):
    """nrf_embed_t"""
    _fields_ = [("mode", C.c_int32), ("multires", C.c_int32), ("multires_views", C.c_int32),
                ("multires_time", C.c_int32)]


_lib = None
ABI_VERSION = 230     # include/nrf.h NRF_VERSION

_SIGS = {
    "nrf_version": (C.c_int, []),
    "nrf_last_error_string": (C.c_char_p, []),
    "nrf_sample_coarse": (C.c_int, [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]),
    "nrf_points": (C.c_int, [vp, C.c_int64, C.c_int32, vp, C.c_int32, vp, vp]),
    "nrf_pe_fwd": (C.c_int, [vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, vp, vp]),
    "nrf_pe_bwd": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, vp, vp]),
    "nrf_mlp_packed_floats": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_mlp_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_query_fwd": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32, vp, vp]),
    "nrf_train_acts_floats": (C.c_int64, [C.POINTER(NrfNet), C.c_int64]),
    "nrf_train_grads_floats": (C.c_int64, [C.POINTER(NrfNet), C.c_int64]),
    "nrf_train_slab_floats": (C.c_int64, [C.POINTER(NrfNet), C.c_int64]),
    "nrf_query_fwd_train": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32, vp,
                                      vp, vp]),
    "nrf_query_bwd": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, C.c_int64,
                                C.POINTER(NrfNet), vp]),
    "nrf_query_bwd_static": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.POINTER(NrfNet),
                                       vp]),
    "nrf_dnerf_time_packed_floats": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_dnerf_pack_time_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_dnerf_query_fwd": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp,
                                      C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp, vp, vp]),
    "nrf_mlp_bf16_packed_bytes": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_mlp_bf16_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_query_fwd_bf16": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32, vp, vp]),
    "nrf_mlp_f16_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_query_fwd_f16": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32, vp, vp]),
    "nrf_query_fwd_bf16x3": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32, vp, vp]),
    "nrf_query_fwd_train_bf16x3": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32, vp,
                                             vp, vp]),
    "nrf_query_bwd_bf16x3": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, vp, C.c_int64,
                                       C.POINTER(NrfNet), vp, vp, vp]),
    "nrf_query_fwd_train_bf16x3_a16": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32, vp,
                                                 vp, vp]),
    "nrf_query_bwd_bf16x3_a16": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, vp, C.c_int64,
                                           C.POINTER(NrfNet), vp]),
    "nrf_query_bwd_bf16x3_w16": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, vp, C.c_int64,
                                           C.POINTER(NrfNet), vp]),
    "nrf_mlp_bf16x3_bwd_packed_bytes": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_mlp_bf16x3_bwd_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_mlp_bf16x3_packed_bytes": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_mlp_bf16x3_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_dnerf_bf16_packed_bytes": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_dnerf_bf16_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_dnerf_query_fwd_bf16": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp,
                                           C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp, vp, vp]),
    "nrf_dnerf_bf16x3_packed_bytes": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_dnerf_bf16x3_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_dnerf_query_fwd_bf16x3": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp,
                                             C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp, vp, vp]),
    "nrf_query_bwd_input": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, C.c_int64,
                                      C.POINTER(NrfNet), vp, vp, vp]),
    "nrf_dnerf_time_fwd_train": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, C.c_int64, C.c_int32,
                                           C.c_int32, vp, vp, vp]),
    "nrf_dnerf_time_bwd": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, C.c_int64,
                                     C.POINTER(NrfNet), vp]),
    "nrf_dnerf_time_bwd_bf16x3": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, vp, C.c_int64,
                                            C.POINTER(NrfNet), vp]),
    "nrf_dnerf_straight_packed_floats": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_dnerf_straight_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_dnerf_straight_query_fwd": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, C.c_int64, C.c_int32,
                                               C.c_int32, vp, vp]),
    "nrf_dnerf_straight_fwd_train": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, C.c_int64, C.c_int32,
                                               C.c_int32, vp, vp, vp]),
    "nrf_dnerf_straight_bwd": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, vp, C.c_int64,
                                         C.POINTER(NrfNet), vp]),
    "nrf_dnerf_straight_bf16x3_packed_bytes": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_dnerf_straight_bf16x3_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_dnerf_straight_query_fwd_bf16x3": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, C.c_int64,
                                                      C.c_int32, C.c_int32, vp, vp]),
    "nrf_dnerf_straight_bf16_packed_bytes": (C.c_int64, [C.POINTER(NrfNet), C.POINTER(NrfEmbed)]),
    "nrf_dnerf_straight_bf16_pack_weights": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp]),
    "nrf_dnerf_straight_query_fwd_bf16": (C.c_int, [C.POINTER(NrfNet), C.POINTER(NrfEmbed), vp, vp, vp, vp, C.c_int64,
                                                    C.c_int32, C.c_int32, vp, vp]),
    "nrf_gen_rays": (C.c_int, [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_int32, vp,
                               C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_float,
                               C.c_float, vp, vp]),
    "nrf_gather_ray_batch": (C.c_int, [vp, C.c_int64, vp, C.c_int32, vp, C.c_int32, C.c_int64, vp, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, vp, vp, vp]),
    "nrf_adam_step": (C.c_int, [C.c_int32, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                vp, vp]),
    "nrf_composite_fwd": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, vp, C.c_int32, vp, vp, vp, vp, vp,
                                    vp]),
    "nrf_composite_bwd": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, vp, C.c_int32, vp, vp, vp, vp, vp,
                                    vp]),
    "nrf_sample_fine": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]),
    "nrf_sample_pdf": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C nerf_amd/csrc` "
                               "(or __graft_entry__.build()); nerf_amd has no CPU fallback")
        # torch bundles its own HIP runtime (libamdhip64); it must be the one already mapped when
        # libnrf.so resolves its dependency, or the process ends up with two runtimes and two device states
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.nrf_version() != ABI_VERSION:      # a stale libnrf.so would read the structs below with another layout
            raise RuntimeError(f"{LIB_PATH} reports ABI version {l.nrf_version()}, this binding is written against "
                               f"{ABI_VERSION} (include/nrf.h NRF_VERSION): rebuild with `make -C nerf_amd/csrc`")
        _lib = l
    return _lib


def exported_symbols():
    return list(_SIGS)


def check(status, what):
    if status != 0:
        msg = lib().nrf_last_error_string().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")
