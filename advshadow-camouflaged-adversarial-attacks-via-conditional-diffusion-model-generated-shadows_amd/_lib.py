"""ctypes binding of ``libadvshadow_hip.so`` (the C ABI declared in ``include/advshadow.h``).

There is no CPU fallback: if the shared library has not been built (``__graft_entry__.build()``
or ``make -C csrc``) importing a kernel raises, and every call checks its status code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADVS_LIB_PATH: load another BUILD of the same library (A/B runs of tools/); never a different implementation
LIB_PATH = os.environ.get("ADVS_LIB_PATH") or os.path.join(_HERE, "libadvshadow_hip.so")

F32, BF16, F16 = 0, 1, 2
ACT = {"none": 0, None: 0, "relu": 1, "silu": 2, "gelu": 3, "relu6": 4, "lrelu": 5, "lrelu001": 6, "sigmoid": 7}
GN_RESIDUAL_AFTER_ACT = 0x100      # include/advshadow.h

vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class ConvArgs(C.Structure):
    """Mirror of ``advs_conv_args``."""
    _fields_ = [("x1", vp), ("x2", vp), ("w", vp), ("bias", vp), ("temb", vp), ("residual", vp), ("y", vp),
                ("b", i32), ("h", i32), ("w_", i32), ("c1", i32), ("c2", i32), ("cout", i32),
                ("ksize", i32), ("stride", i32), ("pad", i32), ("upsample", i32),
                ("act", i32), ("dtype", i32), ("temb_stride", i32), ("tile", i32), ("stats", vp), ("stats_rows", i32), ("e1", vp), ("e2", vp), ("ce1", i32), ("ce2", i32),
                ("ld1", i32), ("ld2", i32), ("relu_mask", vp), ("norm", vp)]


class UNetConfig(C.Structure):
    """include/advshadow.h: advs_unet_config (diff_model.UNetModel's constructor arguments)."""
    _fields_ = [("in_channels", i32), ("model_channels", i32), ("out_channels", i32), ("num_res_blocks", i32),
                ("n_attention_resolutions", i32), ("attention_resolutions", i32 * 8),
                ("n_channel_mult", i32), ("channel_mult", i32 * 8), ("num_heads", i32), ("dtype", i32)]


# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "advs_init": [],
    "advs_abi_version": [],
    "advs_pack_conv_weight": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_nchw_f32_to_nhwc": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_nhwc_to_nchw_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_conv2d": [C.POINTER(ConvArgs), vp],
    "advs_conv_set_tile": [i32],
    "advs_conv_resolve_tile": [C.POINTER(ConvArgs)],
    "advs_conv_tile_rows": [i32],
    "advs_groupnorm_stats": [vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_groupnorm_affine_stats": [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_conv3x3_first": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "advs_conv3x3_first_stats": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "advs_conv_first_stats_rows": [i32, i32, i32, i32, i32],
    "advs_conv_last": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_groupnorm": [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_maxpool2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_concat_upsample2x": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "advs_concat_nearest2x": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "advs_layernorm": [vp, vp, vp, vp, C.c_longlong, i32, f32, i32, vp],
    "advs_attention_masked": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_attention_bias": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_window_shift": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_cls_mean_rows_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_dwconv2d": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_dwconv2d_act": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_scale_channels": [vp, vp, vp, i32, i32, i32, i32, vp],
    "advs_space_to_depth2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_patchify": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "advs_patchify_padded": [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_vit_assemble": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_gather_rows_f32": [vp, vp, i32, C.c_longlong, i32, i32, vp],
    "advs_attention": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_linear_f32": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_timestep_embedding": [vp, vp, i32, i32, vp, vp, vp, i32, vp],
    "advs_ddim_step": [vp, vp, vp, f32, vp, vp, vp, i32, vp, vp, i32, sz, i32, vp],
    "advs_ddpm_step": [vp, vp, vp, f32, vp, vp, vp, i32, vp, vp, i32, sz, vp],
    "advs_ddpm_posterior_step": [vp, vp, vp, vp, vp, i32, vp, vp, i32, sz, i32, vp],
    "advs_plms_combine": [vp, vp, f32, vp, vp, vp, vp, i32, vp, vp, sz, vp],
    "advs_to_uint8": [vp, vp, sz, i32, vp],
    "advs_unit_to_uint8": [vp, vp, sz, vp],
    "advs_apply_shadow": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, C.POINTER(f32), i32, vp],
    "advs_apply_shadow_parts": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, C.POINTER(f32), i32, vp],
    "advs_blend_mask_clamp01": [vp, vp, vp, vp, C.c_longlong, vp],
    "advs_composite_u8": [vp, vp, vp, vp, sz, i32, f32, vp],
    "advs_composite_u8_masks": [vp, vp, vp, vp, i32, vp, sz, i32, f32, vp],
    "advs_mask_contours": [vp, i32, i32, i32, vp, vp, vp, i32, vp],
    "advs_resample_u8": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_u8hwc_to_f32nchw": [vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "advs_u8_nchw_to_hwc": [vp, vp, i32, i32, i32, i32, vp],
    "advs_psnr_ssim": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_jpeg_roundtrip_u8": [vp, vp, vp, i32, i32, i32, i32, vp],
    "advs_argmax_rows": [vp, vp, i32, i32, vp],
    "advs_conv_stem": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_im2col_nchw": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_col2im_nchw": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_maxpool3x3s2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_global_avgpool": [vp, vp, i32, i32, i32, i32, vp],
    "advs_softmax_ce_grad": [vp, vp, vp, i32, i32, f32, vp],
    "advs_relu_bwd": [vp, vp, vp, vp, C.c_longlong, i32, vp],
    "advs_zero_insert2x": [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_gather2x2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_depth_to_space2_relu": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_avgpool_bwd_relu": [vp, vp, vp, i32, i32, i32, i32, vp],
    "advs_maxpool3x3s2_bwd_relu": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_maxpool2_bwd_relu": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_conv_stem_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_iga_step": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp],
    "advs_perturb_clamp01": [vp, vp, vp, C.c_longlong, vp],
    "advs_lerp_stack": [vp, vp, vp, i32, C.c_longlong, vp],
    "advs_graph_begin": [vp],
    "advs_graph_end": [vp, C.POINTER(vp)],
    "advs_graph_launch": [vp, vp],
    "advs_graph_destroy": [vp],
    "advs_event_create": [C.POINTER(vp)],
    "advs_layernorm_bwd": [vp, vp, vp, vp, vp, C.c_longlong, i32, f32, i32, vp],
    "advs_gelu": [vp, vp, C.c_longlong, i32, vp],
    "advs_gelu_bwd": [vp, vp, vp, C.c_longlong, i32, vp],
    "advs_attention_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_scatter_row0": [vp, vp, i32, C.c_longlong, i32, i32, vp],
    "advs_scatter_cls_mean": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_attention_bias_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_dwconv2d_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "advs_depth_to_space2": [vp, vp, i32, i32, i32, i32, i32, vp],
    "advs_avgpool_bwd": [vp, vp, i32, i32, i32, i32, vp],
    "advs_silu": [vp, vp, vp, C.c_longlong, i32, vp],
    "advs_silu_bwd": [vp, vp, vp, C.c_longlong, i32, vp],
    "advs_dwconv2d_bwd_strided": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_channel_dot": [vp, vp, vp, i32, i32, i32, i32, vp],
    "advs_sigmoid_gate_bwd": [vp, vp, vp, C.c_longlong, vp],
    "advs_se_scale_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "advs_unpatchify_padded": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "advs_event_record": [vp, vp],
    "advs_event_elapsed_ms": [vp, vp, C.POINTER(f32)],
    "advs_event_destroy": [vp],
    "advs_stream_sync": [vp],
    # handle-level entry points (csrc/unet_handle.hip; handle.py is their ctypes view)
    "advs_unet_create": [C.POINTER(UNetConfig), C.POINTER(vp)],
    "advs_unet_param_count": [vp],
    "advs_unet_param_name": [vp, i32, C.c_char_p, i32, C.POINTER(C.c_longlong)],
    "advs_unet_set_param": [vp, C.c_char_p, vp, C.c_longlong],
    "advs_unet_plan": [vp, i32, i32, i32, vp],
    "advs_unet_forward": [vp, vp, vp, vp],
    "advs_ddim_tables": [i32, i32, i32, i32, f32, vp, vp, C.POINTER(i32)],
    "advs_ddim_run": [vp, vp, vp, vp, i32, i32],
    "advs_resize_tables": [i32, i32, vp, vp, C.POINTER(i32)],
    "advs_resnet50_create": [i32, i32, C.POINTER(vp)],
    "advs_resnet50_param_count": [vp],
    "advs_resnet50_param_name": [vp, i32, C.c_char_p, i32, C.POINTER(C.c_longlong)],
    "advs_resnet50_set_param": [vp, C.c_char_p, vp, C.c_longlong],
    "advs_resnet50_plan": [vp, i32, i32, i32, vp],
    "advs_resnet50_forward": [vp, vp, vp],
    "advs_resnet50_eval_u8": [vp, vp, vp],
}
_RESTYPES = {"advs_last_error": C.c_char_p, "advs_groupnorm_scratch_bytes": sz, "advs_jpeg_scratch_bytes": sz,
             "advs_mask_contours_work_bytes": sz, "advs_attention_bwd_scratch_bytes": sz, "advs_unet_destroy": None, "advs_resnet50_destroy": None}
_EXTRA = {"advs_last_error": [], "advs_groupnorm_scratch_bytes": [i32, i32], "advs_jpeg_scratch_bytes": [i32, i32, i32],
          "advs_mask_contours_work_bytes": [i32, i32, i32], "advs_attention_bwd_scratch_bytes": [i32, i32, i32], "advs_unet_destroy": [vp], "advs_resnet50_destroy": [vp]}

_lib = None


class AdvsError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # torch first: its bundled libamdhip64 (SONAME libamdhip64.so.7) must be the HIP runtime already in
        # the process when this library's NEEDED libamdhip64.so.7 is resolved, otherwise the system copy is
        # loaded beside torch's and streams / device pointers would cross two runtimes.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise AdvsError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                            f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
        if os.environ.get("ADVS_LIB_PATH"):
            import sys
            print(f"advshadow_amd: loading the library named by ADVS_LIB_PATH: {LIB_PATH}", file=sys.stderr)
        lib = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for name, args in _EXTRA.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPES[name]
        _lib = lib
    return _lib


def exported_symbols():
    """Every entry point ``include/advshadow.h`` declares (used by the CPU-side ABI test)."""
    return list(SIGNATURES) + list(_EXTRA)


def check(rc, what=""):
    if rc != 0:
        msg = load().advs_last_error()
        raise AdvsError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


_initialised = set()


def init_device():
    """Allocate the library's zero page on the current device (idempotent)."""
    import torch
    if not torch.cuda.is_available():
        raise AdvsError("no GPU visible: the AdvShadow HIP path needs an MI355X (there is no CPU fallback)")
    dev = torch.cuda.current_device()
    if dev not in _initialised:
        check(load().advs_init(), "advs_init")
        _initialised.add(dev)
