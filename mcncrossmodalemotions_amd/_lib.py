"""ctypes binding of libxmodal_hip.so (the C ABI in include/xmodal.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing
an operator raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes as C
import os

# torch FIRST: its wheel bundles its own ROCm runtime (libamdhip64 / libhsa-runtime64 / librccl under
# torch/lib).  libxmodal_hip.so is linked against the same sonames; once torch's copies are mapped the
# loader reuses them and the process has ONE HIP runtime.  Loaded the other way round, the library pulls in
# /opt/rocm/lib's copies, torch then maps its own next to them, and the process aborts in the runtimes'
# static destructors at exit ("double free or corruption").
import torch  # noqa: F401  (side effect: loads torch/lib/libamdhip64.so and friends)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("XM_LIB_PATH") or os.path.join(_HERE, "libxmodal_hip.so")   # XM_LIB_PATH: A/B experiments

c_fp = C.c_void_p  # device pointers travel as raw addresses
_i, _f, _sz, _vp = C.c_int, C.c_float, C.c_size_t, C.c_void_p

# name -> argtypes; every function returns int (status) unless listed in _RESTYPES
SIGNATURES = {
    "xm_version": [],
    "xm_last_error": [],
    "xm_tune_load": [C.c_char_p],
    "xm_tune_save": [C.c_char_p],
    "xm_tune_entries": [C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "xm_set_exec_hint": [C.c_uint],
    "xm_get_exec_hint": [],
    "xm_workspace_reserve": [_sz],
    "xm_workspace_bytes": [],
    "xm_workspace_reserve_stream": [_sz, _vp],
    "xm_workspace_generation": [],
    "xm_device_alloc": [C.POINTER(C.c_void_p), _sz],
    "xm_device_free": [_vp],
    "xm_device_upload": [_vp, _vp, _sz],
    "xm_device_download": [_vp, _vp, _sz],
    "xm_device_synchronize": [],
    "xm_out_size": [_i] * 6,
    "xm_nnconv_forward": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp, c_fp] + [_i] * 8 + [_vp],
    "xm_nnconv_forward_moments": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp, c_fp] + [_i] * 8 + [_f, c_fp, _vp],
    "xm_nnconv_forward_gated": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp, c_fp] + [_i] * 8 +
                               [c_fp, c_fp, c_fp, c_fp, _i, _vp],
    "xm_nnconv_forward_fused": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp, c_fp] + [_i] * 8 +
                               [c_fp, c_fp, c_fp, _i, _vp],
    "xm_nnconv_backward": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, c_fp] +
                          [_i] * 8 + [_vp],
    "xm_nnconv_backward_accum": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, c_fp] +
                                [_i] * 8 + [c_fp, _vp],
    "xm_nnconv_prepare_backward": [_i] * 4 + [c_fp] + [_i] * 12 + [_vp],
    "xm_nnconv_backward_filter_bnrelupool": [c_fp] + [_i] * 16 + [c_fp, c_fp, c_fp, c_fp] + [_i] * 9 +
                                            [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _vp],
    "xm_stem_gram": [c_fp] + [_i] * 11 + [_vp, _vp],
    "xm_stem_gram_moments": [_vp, c_fp, c_fp, _i, _i, _i, _f, c_fp, _vp],
    "xm_nnconv_backward_filter_bnrelupool_gram": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp] + [_i] * 8 +
                                                 [c_fp, c_fp] + [_i] * 9 + [c_fp, c_fp, c_fp, _vp, c_fp, c_fp, c_fp, c_fp, _vp],
    "xm_nnconv_bnorm_relu_pool_forward": [c_fp] + [_i] * 4 + [c_fp] + [_i] * 4 + [c_fp] + [_i] * 8 + [c_fp, c_fp, _f, c_fp] +
                                         [_i] * 8 + [_vp, c_fp, c_fp, c_fp, _vp],
    "xm_params_changed": [],
    "xm_nnpool_forward": [c_fp] + [_i] * 13 + [c_fp, _vp],
    "xm_nnpool_backward": [c_fp] + [_i] * 13 + [c_fp, c_fp, _vp],
    "xm_nnpool_forward_argmax": [c_fp] + [_i] * 12 + [c_fp, c_fp, _vp],
    "xm_nnpool_backward_argmax": [c_fp] + [_i] * 12 + [c_fp, c_fp, _vp],
    "xm_nnpool_global_avg_backward_accum": [c_fp, c_fp, c_fp, _i, _i, _i, _i, _vp],
    "xm_nnbnorm_forward": [c_fp] + [_i] * 4 + [c_fp, c_fp, _f, c_fp, c_fp, c_fp, _vp],
    "xm_nnbnorm_forward_fused": [c_fp] + [_i] * 4 + [c_fp, c_fp, _f, c_fp, c_fp, c_fp, _i, _vp],
    "xm_nnbnorm_backward": [c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, _f, c_fp, c_fp, c_fp, c_fp, c_fp,
                                               _vp],
    "xm_nnbnorm_backward_fused": [c_fp, c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, _f, c_fp, c_fp, c_fp,
                                                             c_fp, c_fp, _i, _vp],
    "xm_nnbnorm_backward_dxsum": [c_fp, c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, _f, c_fp, c_fp, c_fp,
                                                             c_fp, c_fp, c_fp, _i, _vp],
    "xm_nnbnorm_relu_pool_forward": [c_fp] + [_i] * 4 + [c_fp, c_fp, _f, c_fp] + [_i] * 8 +
                                    [c_fp, c_fp, c_fp, _vp],
    "xm_nnbnorm_relu_pool_backward": [c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, _i] + [_i] * 8 +
                                     [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _vp],
    "xm_nndropout_forward": [c_fp, _sz, _f, C.c_ulonglong, C.c_ulonglong, c_fp, c_fp, _vp],
    "xm_nndropout_apply": [c_fp, c_fp, _sz, c_fp, _vp],
    "xm_nnrelu": [c_fp, _sz, _f, c_fp, c_fp, _vp],
    "xm_nnsigmoid": [c_fp, _sz, c_fp, c_fp, _vp],
    "xm_sum2": [c_fp, c_fp, _sz, _i, c_fp, _vp],
    "xm_scale_axpy": [c_fp, _i, _i, c_fp, c_fp, _i, c_fp, _vp],
    "xm_scale_backward": [c_fp, _i, _i, c_fp, c_fp, c_fp, c_fp, _vp],
    "xm_se_squeeze_bn": [c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, c_fp, _vp],
    "xm_scale_axpy_bn": [c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, c_fp, c_fp, _i, c_fp, _vp],
    "xm_se_tail_backward_reduce": [c_fp, c_fp, c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, c_fp, c_fp, _vp],
    "xm_se_tail_backward_apply": [c_fp, c_fp, c_fp] + [_i] * 4 + [c_fp, c_fp, c_fp, c_fp, _i, c_fp, c_fp, c_fp, c_fp, c_fp, _vp],
    "xm_nnsoftmaxt": [c_fp, _i, _i, _i, _f, c_fp, _vp],
    "xm_nnsoftmaxt_backward": [c_fp, c_fp, _i, _i, _i, _f, c_fp, _vp],
    "xm_nnregloss": [c_fp, c_fp, _i, _i, _i, _f, c_fp, c_fp, c_fp, _vp],
    "xm_nnsoftmaxceloss": [c_fp, c_fp, _i, _i, _f, _i, c_fp, c_fp, c_fp, _vp],
    "xm_nnloss": [c_fp, c_fp, _i, _i, _i, c_fp, c_fp, _vp],
    "xm_sgd_update": [c_fp, c_fp, c_fp, _sz, _f, _f, _f, _f, _vp],
    "xm_average_update": [c_fp, c_fp, _sz, _f, _f, _vp],
    "xm_comm_unique_id": [_vp],
    "xm_comm_init": [_vp, _i, _i],
    "xm_allreduce_sum_f32": [c_fp, _sz, _vp],
    "xm_parserv_push": [c_fp, _sz, _vp],
    "xm_parserv_sync": [_vp],
    "xm_comm_count": [C.POINTER(C.c_int)],
    "xm_scale_f32": [c_fp, _sz, _f, _vp],
    "xm_comm_destroy": [],
    "xm_spec_rownorm": [c_fp, _i, _i, _i, c_fp, _vp],
    "xm_spec_magnitude": [c_fp, _i, _i, _i, c_fp, _vp],
    "xm_resample": [c_fp, _i, c_fp, _i, _i, _i, _i, c_fp, _i, _vp],
    "xm_aggregate_logits": [c_fp, _i, _i, c_fp, c_fp, _i, _i, c_fp, c_fp, _vp],
    "xm_max_label": [c_fp, _i, _i, c_fp, _vp],
    "xm_class_stats": [c_fp, c_fp, _i, _i, c_fp, c_fp, _vp],
    "xm_normalize_face": [c_fp, _i, _i, _i, C.POINTER(C.c_float), c_fp, _vp],
    "xm_crop_resize_face": [c_fp, _i, _i, _i, _f, _i, _i, C.POINTER(C.c_float), c_fp, _vp],
}
_RESTYPES = {"xm_get_exec_hint": C.c_uint, "xm_last_error": C.c_char_p, "xm_workspace_bytes": C.c_size_t,
             "xm_workspace_generation": C.c_ulonglong}
# test / tools switches and measurement hooks (include/xmodal_prof.h; not part of include/xmodal.h)
_DEBUG = {"xm_debug_set": [C.c_char_p, _i], "xm_debug_get": [C.c_char_p],
          "xm_prof_enable": [_i],
          "xm_prof_collect": [_i, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                              C.POINTER(C.c_longlong)],
          "xm_prof_collect_bytes": [_i, C.POINTER(C.c_int), C.POINTER(C.c_double)],
          "xm_prof_kernel_name": [_i, C.c_char_p, _i]}
# the library exports ONE switch entry (xm_debug_set); tests and tools keep calling L.xm_debug_force_<what>(v) through
# these shims (name -> key)
_SWITCHES = {"xm_debug_force_conv_cfg": "conv_cfg", "xm_debug_force_conv_splits": "conv_splits",
             "xm_debug_force_conv_halo": "conv_halo", "xm_debug_force_conv_stem": "conv_stem",
             "xm_debug_force_conv_stem3": "conv_stem3", "xm_debug_force_wgrad_patch": "wgrad_patch",
             "xm_debug_force_wgrad_patch_s2": "wgrad_patch_s2", "xm_debug_force_dgrad_s2": "dgrad_s2"}

_lib = None


class XmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("xmodal error %d: %s" % (code, msg))
        self.code = code


def load():
    """dlopen the HIP library; raises if it is absent (no CPU fallback, by design)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "%s not found: build it with `python -m mcncrossmodalemotions_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % SO_PATH)
    lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    for name, args in list(SIGNATURES.items()) + list(_DEBUG.items()):
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    for name, key in _SWITCHES.items():
        setattr(lib, name, (lambda k: (lambda v: lib.xm_debug_set(k, int(v))))(key.encode()))
    lib.xm_debug_num_conv_cfgs = lambda: lib.xm_debug_get(b"num_conv_cfgs")
    lib.xm_debug_comm_force_single = lambda v: 0 if lib.xm_debug_set(b"comm_single", int(v)) > -2 else 1
    if hasattr(lib, "xm_debug_conv_cycles"):          # tools build only (XM_DEBUG_CYCLES=1)
        lib.xm_debug_conv_cycles.argtypes = [_i, C.POINTER(C.c_ulonglong), _i]
        lib.xm_debug_conv_cycles.restype = C.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise XmError(rc, load().xm_last_error().decode("utf-8", "replace"))
