"""ctypes binding of libwsi_hip.so (C ABI: include/wsi_hip.h).

The product path has no CPU fallback: if the library is missing, ``load()`` raises and every caller
fails loudly.  ``build()`` compiles the library in-tree with hipcc for gfx950 (works without a GPU).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libwsi_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


class WsiTrunkWeights(C.Structure):
    _fields_ = [
        ('stem_w', C.c_void_p), ('stem_b', C.c_void_p),
        ('stem_w_u8', C.c_void_p), ('stem_b_u8', C.c_void_p), ('norm', C.c_float * 6),
        ('conv_w', C.c_void_p * 16), ('conv_b', C.c_void_p * 16),
        ('down_w', C.c_void_p * 3), ('down_b', C.c_void_p * 3),
        ('head_w', C.c_void_p), ('head_b', C.c_void_p), ('head_k', C.c_int),
        ('planes', C.c_int),
    ]


class WsiUnetDecoderWeights(C.Structure):
    _fields_ = [
        ('conv_w', C.c_void_p * 10), ('conv_b', C.c_void_p * 10), ('cin', C.c_int * 10), ('cout', C.c_int * 10),
        ('head_w', C.c_void_p), ('head_b', C.c_void_p), ('head_cin', C.c_int), ('classes', C.c_int), ('tail_w', C.c_void_p),
    ]


_vp, _i, _ll, _sz, _f, _d = C.c_void_p, C.c_int, C.c_longlong, C.c_size_t, C.c_float, C.c_double
# name -> (restype, argtypes); must list every symbol include/wsi_hip.h declares
SIGNATURES = {
    'wsi_hip_abi_version': (_i, []),
    'wsi_pf_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'wsi_pf_pixel_index': (_ll, [_i, _i, _i, _i, _i]),
    'wsi_prepack_conv_bytes': (_sz, [_i, _i, _i, _i]),
    'wsi_prepack_conv': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp]),
    'wsi_unet_tail_prepack_bytes': (_sz, []),
    'wsi_unet_tail_prepack': (_i, [_vp] * 10 + [_f, _vp, _vp, _i, _i, _i, _vp]),
    'wsi_prepack_stem_bytes': (_sz, [_i]),
    'wsi_prepack_stem': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp]),
    'wsi_prepack_stem_u8': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp]),
    'wsi_normalize_u8_lut': (_i, [_vp, _vp, _vp]),
    'wsi_stem_conv7x7_bn_relu_maxpool': (_i, [_vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    'wsi_stem_set_mode': (_i, [_i, _i]),
    'wsi_conv3x3_bn_act': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'wsi_conv3x3_bn_act_cfg': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'wsi_conv3x3s2_ds_fused': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'wsi_pf_split_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'wsi_conv3x3_bn_act_split': (_i, [_vp] * 5 + [_i] * 7 + [_vp]),
    'wsi_conv3x3_up_concat_bn_act': (_i, [_vp] * 5 + [_i] * 8 + [_vp]),
    'wsi_conv3x3s2_ds_fused_split': (_i, [_vp] * 7 + [_i] * 6 + [_vp]),
    'wsi_conv_set_mode': (_i, [_i]),
    'wsi_conv1x1_bn': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'wsi_avgpool_fc': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    'wsi_linear': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'wsi_pf_pack': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'wsi_pf_unpack': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'wsi_trunk_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'wsi_trunk_workspace_init': (_i, [_vp, _i, _i, _i, _i, _vp]),
    'wsi_trunk_workspace_release': (_i, [_vp]),
    'wsi_trunk_forward': (_i, [C.POINTER(WsiTrunkWeights), _vp, _vp, _ll, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    'wsi_trunk_forward_tap': (_i, [C.POINTER(WsiTrunkWeights), _vp, _vp, _ll, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    'wsi_trunk_set_chunks': (_i, [_i, _i]),
    'wsi_prof_begin': (_i, [_i]),
    'wsi_prof_end': (_i, [_vp, _vp, _vp, _i]),
    'wsi_tile_gather': (_i, [_vp, _ll, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'wsi_stitch_add': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    'wsi_stitch_add_dense': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    'wsi_softmax_threshold_argmax': (_i, [_vp, _i, _ll, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'wsi_ring_create': (_i, [_vp, _i, _sz]),
    'wsi_ring_host_slot': (_vp, [_vp, _i]),
    'wsi_ring_wait_slot': (_i, [_vp, _i]),
    'wsi_ring_submit': (_i, [_vp, _i, _i, _i, _i, _ll, _vp, _ll]),
    'wsi_ring_fence': (_i, [_vp, _vp]),
    'wsi_ring_acquire': (_i, [_vp, _vp]),
    'wsi_ring_device': (_i, [_vp]),
    'wsi_ring_drain': (_i, [_vp]),
    'wsi_ring_destroy': (None, [_vp]),
    'wsi_resample_plan_create': (_i, [_vp, _i, _i, _i, _i]),
    'wsi_resample_plan_destroy': (None, [_vp]),
    'wsi_resample_scratch_bytes': (_sz, [_vp, _i]),
    'wsi_resample_tiles': (_i, [_vp, _vp, _ll, _i, _i, _vp, _i, _vp, _vp, _vp]),
    'wsi_find_nuclei_hsv': (_i, [_vp, _ll, _i, C.c_double, _vp, _vp]),
    'wsi_connected_components_scratch_bytes': (_sz, [_i, _i]),
    'wsi_connected_components': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    'wsi_kmeans_points': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    'wsi_kmeans_seed_farthest': (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    'wsi_find_nuclei_lab': (_i, [_vp, _ll, _i, _d, _vp, _vp, _vp]),
    'wsi_fill_holes_scratch_bytes': (_sz, [_i, _i]),
    'wsi_fill_holes': (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    'wsi_slic_scratch_bytes': (_sz, [_i, _i, _i]),
    'wsi_slic': (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _d, _d, _i, _vp, _vp, _vp]),
    'wsi_tile_grid_candidates': (_ll, [_i, _i, _i, _i, _i, _i]),
    'wsi_tile_grid_scratch_bytes': (_sz, [_ll]),
    'wsi_tile_grid': (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _i, C.c_double, C.c_double, _vp, _vp, _vp, _vp]),
    'wsi_exponent_span': (_i, [_vp, _ll, _vp, _vp]),
    'wsi_paint_regions': (_i, [_vp, _vp, _ll, _vp, _vp, _vp, _ll, _vp]),
    'wsi_resize_bilinear_f64': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    'wsi_argmax_classes': (_i, [_vp, _i, _ll, _vp, _vp]),
    'wsi_morph_rect': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'wsi_bwperim': (_i, [_vp, _vp, _i, _i, _vp]),
    'wsi_tumor_bed_workspace_bytes': (_sz, [_i, _i]),
    'wsi_convex_hull_image': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'wsi_tumor_bed': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'wsi_hull_polygon': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    'wsi_mask_iou_counts': (_i, [_vp, _vp, _ll, _vp, _vp]),
    'wsi_score_counts': (_i, [_vp, _vp, _vp, _ll, _vp, _vp]),
    'wsi_esp': (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    'wsi_unet_workspace_bytes': (_sz, [C.POINTER(WsiUnetDecoderWeights), _i, _i, _i, _i]),
    'wsi_unet_workspace_init': (_i, [C.POINTER(WsiUnetDecoderWeights), _vp, _i, _i, _i, _i, _vp]),
    'wsi_unet_forward': (_i, [C.POINTER(WsiTrunkWeights), C.POINTER(WsiUnetDecoderWeights), _vp, _vp, _ll, _i, _i, _vp, _vp, _i, _i, _i,
                              _vp, _i, _vp, C.POINTER(C.c_void_p * 5), _vp]),
    'wsi_resize_nearest_f32': (_i, [_vp, _ll, _i, _i, _vp, _i, _i, _vp]),
    'wsi_unet_decoder': (_i, [C.POINTER(WsiUnetDecoderWeights), C.POINTER(C.c_void_p * 5), _i, _i, _i, _i, _vp, _i, _vp, _vp]),
}


def build(verbose=False):
    """Compile csrc/*.hip into lib/libwsi_hip.so with hipcc --offload-arch=gfx950."""
    res = subprocess.run(['make', '-C', CSRC, '-j8'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode:
        print(res.stdout)
    if res.returncode:
        raise RuntimeError('hipcc build of libwsi_hip.so failed')
    global _lib
    _lib = None
    return LIB_PATH


ABI_VERSION = 6                          # include/wsi_hip.h WSI_HIP_ABI_VERSION (tests/test_capi_symbols.py compares the two)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            '%s not found: the WSI inference path has no CPU fallback - build it first '
            '(python -c "import __graft_entry__ as g; g.build()" or make -C %s)' % (LIB_PATH, CSRC))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lost a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.wsi_hip_abi_version() != ABI_VERSION:
        raise RuntimeError('libwsi_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with code %d (see include/wsi_hip.h error conventions)' % (what, rc))
