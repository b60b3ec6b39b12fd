"""ctypes binding of libhoig_hip.so (the C ABI declared in include/hoig_kernels.h).

There is NO fallback: if the shared library is missing, or a symbol the header
declares is absent, importing this module raises.  Every product op in
``hoig_amd.ops`` goes through here.
"""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', '_build', 'libhoig_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'hoig_kernels.h')

OK, EINVAL, ELAUNCH, EUNSUPPORTED = 0, -1, -2, -3
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16X2, PREC_F16F6 = 0, 1, 2, 3, 4
LOSS_L1, LOSS_MSE, LOSS_BCE = 0, 1, 2
_ERR = {EINVAL: 'invalid argument', ELAUNCH: 'kernel launch failed', EUNSUPPORTED: 'unsupported shape'}


class HoigKernelError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('B', 'Hi', 'Wi', 'Ci', 'Ho', 'Wo', 'Co', 'R', 'S', 'stride', 'pad',
                                              'transposed', 'act')] + \
               [('slope', ctypes.c_float), ('precision', ctypes.c_int32)]


def build(verbose=False):
    """Compile every HIP source for gfx950 into hoig_amd/csrc/_build/libhoig_hip.so."""
    script = os.path.join(_HERE, 'csrc', 'build.sh')
    out = subprocess.run(['bash', script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError('hipcc build of libhoig_hip.so failed')
    return LIB_PATH


def declared_symbols():
    """Entry points the header declares (used by the CPU test that checks the library exports all of them)."""
    text = open(HEADER_PATH).read()
    return sorted(set(re.findall(r'\b(hoig_[a-z0-9_]+)\s*\(', text)) - {'hoig_stream_t'})


_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
_SIGS = {
    'hoig_conv2d_fwd': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_bwd_data': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp],
    'hoig_conv2d_fwd_heads': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp],
    'hoig_conv2d_bwd_weight': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp],
    'hoig_pack_conv_weight_bf16': [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    'hoig_pack_conv_weights_bf16_all': [_vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_fwd_packed': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_bwd_data_packed': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_fwd_packed_stats': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_fwd_stats': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_cat_fwd_packed_stats': [ctypes.POINTER(ConvDesc), _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_inorm_stats_from_sums': [_i, _i, _i, _f, _vp, _vp, _vp, _vp],
    'hoig_conv2d_bwd_data_packed_add': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_pack_conv_weight_f6': [_vp, _i, _i, _i, _vp, _vp, _vp],
    'hoig_conv2d_fwd_f6': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_pack_conv_weights_f6_all': [_vp, _vp, _i, _i64, _vp, _vp, _vp],
    'hoig_conv2d_cat_fwd_f6': [ctypes.POINTER(ConvDesc), _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_fwd_f6_ex': [ctypes.POINTER(ConvDesc), _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    'hoig_conv2d_cat_fwd_packed': [ctypes.POINTER(ConvDesc), _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_cat_bwd_data_packed': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _i, _vp, _vp],
    'hoig_split_planes_bf16': [_vp, _vp, _i64, _i, _vp],
    'hoig_conv2d_bwd_weight_split': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp],
    'hoig_conv2d_bwd_data_packed_split': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_conv2d_fwd_packed_pair': [ctypes.POINTER(ConvDesc)] + [_vp] * 11,
    'hoig_conv2d_bwd_data_packed_split_pair': [ctypes.POINTER(ConvDesc)] + [_vp] * 11,
    'hoig_conv2d_bwd_weight_split_pair': [ctypes.POINTER(ConvDesc)] + [_vp] * 7,
    'hoig_conv2d_cat_bwd_weight': [ctypes.POINTER(ConvDesc), _vp, _i, _vp, _vp, _vp, _vp, _vp],
    'hoig_inorm_stats': [_vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    'hoig_inorm_apply': [_vp, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _i, _vp],
    'hoig_inorm_apply_ld': [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _vp, _vp, _i, _i, _i, _vp],
    'hoig_inorm_bwd_ld': [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    'hoig_inorm_fwd_fused': [_vp, _i, _vp, _vp, _i, _i, _f, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    'hoig_inorm_bwd_fused': [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    'hoig_inorm_bwd_add_ld': [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    'hoig_inorm_bwd_add_ld_split': [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    'hoig_inorm_bwd_fused_add_split': [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    'hoig_inorm_fold': [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp],
    'hoig_conv2d_fwd_packed_normin': [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    'hoig_unsplit_planes_bf16': [_vp, _vp, _i64, _i, _vp],
    'hoig_inorm_bwd_fused_add': [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    'hoig_inorm_bwd': [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    'hoig_replicate_pad_fwd': [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    'hoig_replicate_pad_bwd': [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    'hoig_replicate_pad_bwd_add': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'hoig_attn_pixel_fwd': [_vp] * 10 + [_i, _i, _i, _i, _vp],
    'hoig_attn_pixel_bwd': [_vp] * 10 + [_i, _i, _i, _i, _vp],
    'hoig_attn_build_index': [_vp, _vp, _i, _i, _i, _vp],
    'hoig_attn_src_gather': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'hoig_attn_gs_gather': [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    'hoig_block_extractor_forward': [_vp, _vp, _vp] + [_i] * 7 + [_vp],
    'hoig_block_extractor_backward': [_vp] * 5 + [_i] * 7 + [_vp],
    'hoig_local_attn_reshape_forward': [_vp, _vp] + [_i] * 4 + [_vp],
    'hoig_local_attn_reshape_backward': [_vp, _vp] + [_i] * 4 + [_vp],
    'hoig_grid_sample_fwd': [_vp, _vp, _vp] + [_i] * 6 + [_vp],
    'hoig_grid_sample_bwd': [_vp, _vp, _vp] + [_i] * 6 + [_vp],
    'hoig_resize_bilinear_ac': [_vp, _vp] + [_i] * 6 + [_vp],
    'hoig_resize_nearest': [_vp, _vp] + [_i] * 6 + [_vp],
    'hoig_attn_flow': [_vp, _vp, _i, _i, _vp],
    'hoig_maxpool2_fwd': [_vp, _vp] + [_i] * 4 + [_vp],
    'hoig_maxpool2_bwd': [_vp, _vp, _vp, _vp] + [_i] * 4 + [_vp],
    'hoig_nchw_to_nhwc': [_vp, _vp] + [_i] * 4 + [_vp],
    'hoig_nhwc_to_nchw': [_vp, _vp] + [_i] * 4 + [_vp],
    'hoig_copy_channels': [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp],
    'hoig_cat2_channels': [_vp, _i, _vp, _i, _vp, _i64, _vp],
    'hoig_add': [_vp, _vp, _vp, _i64, _vp],
    'hoig_add_act': [_vp, _vp, _vp, _i, _f, _i64, _vp],
    'hoig_act_bwd': [_vp, _vp, _vp, _i, _f, _i64, _vp],
    'hoig_act_bwd_colsum': [_vp, _vp, _vp, _vp, _i, _f, _i64, _i, _vp],
    'hoig_colsum_accum': [_vp, _vp, _i64, _i, _vp],
    'hoig_compose_fwd': [_vp] * 6 + [_i64, _i, _vp],
    'hoig_compose_bwd': [_vp] * 11 + [_i64, _i, _vp],
    'hoig_loss_fwd_bwd': [_i, _vp, _vp, _f, _f, _vp, _vp, _i64, _vp],
    'hoig_tv_fwd_bwd': [_vp, _f, _f, _vp, _vp, _i, _i, _i, _vp],
    'hoig_loss_accumulate': [_i, _vp, _vp, _f, _f, _vp, _vp, _i64, _vp],
    'hoig_tv_accumulate': [_vp, _f, _f, _vp, _vp, _i, _i, _i, _vp],
    'hoig_sum': [_vp, _vp, _i64, _vp],
    'hoig_sum_scaled': [_vp, _f, _vp, _i64, _vp],
    'hoig_adam_step': [_vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _i, _f, _vp],
    'hoig_adam_tick': [_vp, _vp, _vp],
    'hoig_adam_step_dev': [_vp, _vp, _vp, _vp, _i64, _vp, _f, _vp],
    'hoig_resize_linear_u8': [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp],
    'hoig_warp_affine_u8': [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp],
    'hoig_adam_pack_step': [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    'hoig_stream_create': [ctypes.POINTER(ctypes.c_void_p)],
    'hoig_stream_destroy': [_vp],
    'hoig_stream_scratch_set': [_vp, _vp, _i64],
    'hoig_tensor2im_u8': [_vp, _vp] + [_i] * 6 + [_vp],
    'hoig_prep_texture': [_vp] * 9,
    'hoig_prep_lookup': [_vp] * 5 + [_i] + [_vp] * 8,
    'hoig_prep_assemble': [_i] + [_vp] * 17 + [_i] + [_vp] * 6,
    'hoig_rasterize_fim_wim': [_vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp],
    'hoig_pack_conv_weight_wino': [_vp, _i, _i, _vp, _vp, _vp],
    'hoig_conv2d_fwd_wino': [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_project_faces': [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _f, _f, _vp, _vp],
    'hoig_prep_texture_batched': [_i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_prep_lookup_batched': [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    'hoig_mano_lbs': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'hoig_amd: %s not found. The HIP extension is mandatory (no CPU / eager fallback exists); build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or `bash hoig_amd/csrc/build.sh`.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = args
        fn.restype = ctypes.c_int
    lib.hoig_inorm_workspace_bytes.argtypes = [_i, _i, _i]
    lib.hoig_inorm_workspace_bytes.restype = ctypes.c_int64
    lib.hoig_set_f6_min_tiles.argtypes = [_i]
    lib.hoig_set_f6_min_tiles.restype = ctypes.c_int
    lib.hoig_f6_plane_bytes.argtypes = [_i, _i, _i]
    lib.hoig_f6_plane_bytes.restype = ctypes.c_int64
    lib.hoig_wino_plane_halfs.argtypes = [_i, _i]
    lib.hoig_wino_plane_halfs.restype = ctypes.c_int64
    lib.hoig_attn_index_ints.argtypes = [_i, _i, _i]
    lib.hoig_attn_index_ints.restype = ctypes.c_int64
    lib.hoig_rasterize_workspace_bytes.restype = ctypes.c_size_t
    lib.hoig_rasterize_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.hoig_set_tuning.argtypes = [ctypes.c_char_p, _i]
    lib.hoig_set_tuning.restype = ctypes.c_int
    lib.hoig_stream_scratch_bytes.argtypes = []
    lib.hoig_stream_scratch_bytes.restype = ctypes.c_int64
    lib.hoig_conv2d_bwd_weight_scratch_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.hoig_conv2d_bwd_weight_scratch_bytes.restype = ctypes.c_int64
    lib.hoig_version.argtypes = []
    lib.hoig_version.restype = ctypes.c_char_p
    return lib


lib = _load()


def set_tuning(key, value):
    """hoig_set_tuning: kernel-variant choice `key` -> `value` (returns the previous value)."""
    prev = lib.hoig_set_tuning(key.encode(), int(value))
    if prev < 0:
        raise KeyError('unknown tuning key %r' % key)
    return prev


# HOIG_TUNING="key=value,key=value": variant choices for A/B runs (bench.py and the tools set them through this one variable)
for _kv in filter(None, os.environ.get('HOIG_TUNING', '').split(',')):
    set_tuning(*_kv.split('='))


def check(rc, what):
    if rc != 0:
        raise HoigKernelError('%s failed: %s (code %d)' % (what, _ERR.get(rc, 'unknown'), rc))


def call(name, *args):
    check(getattr(lib, name)(*args), name)
