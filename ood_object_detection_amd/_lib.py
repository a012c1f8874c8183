"""ctypes loader for libeffdet_hip.so (the C ABI declared in include/effdet_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libeffdet_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

_lib = None

c_void_p, c_int, c_ll, c_float, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_double
P = ctypes.POINTER

# name -> (restype, argtypes); must list every symbol include/effdet_hip.h declares
SIGNATURES = {
    'effdet_abi_version': (c_int, []),
    'effdet_last_error': (ctypes.c_char_p, []),
    'effdet_device_error': (c_int, [c_int]),
    'effdet_stem_conv': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int]),
    'effdet_stem_dw_fused': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    'effdet_resize_pad_u8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'effdet_normalize_u8': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll]),
    'effdet_stem_conv_u8': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int]),
    'effdet_stem_dw_fused_u8': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    'effdet_stem_dw_tiles_per_image': (c_int, [c_int, c_int]),
    'effdet_stem_dw_parts': (c_int, [c_int, c_int, c_int, c_int]),
    'effdet_pw_gemm_bn_act': (c_int, [c_void_p, c_int, c_void_p, c_ll, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                      c_int, c_void_p, c_void_p, c_int, c_void_p, c_ll, c_ll]),
    'effdet_pw_gemm_group': (c_int, [c_void_p, c_int, c_int, P(c_void_p), P(c_ll), P(c_int), P(c_void_p), P(c_int), P(c_void_p),
                                     P(c_void_p), c_int, P(c_void_p)]),
    'effdet_dwconv_bn_act': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_dwconv_blocks_per_image': (c_int, [c_int, c_int, c_int]),
    'effdet_mbconv_gated_tiles_per_image': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_mbconv_expand_dw_gated': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_mbconv_expand_dw': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_mbconv_tiles_per_image': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_se_gate': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_int, c_int, c_int]),
    'effdet_maxpool_same': (c_int, [c_void_p, c_int, c_void_p, c_ll, c_void_p, c_ll, c_int, c_int, c_int, c_int]),
    'effdet_sepconv_fused': (c_int, [c_void_p, c_int, c_int, c_int, P(c_int), c_int,
                                     P(c_void_p), P(c_ll), P(c_int), P(c_int), c_int, P(c_float), c_float, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, P(c_int), c_int, c_int, c_int,
                                     P(c_void_p), P(c_ll), c_int, c_int, c_void_p, c_void_p, c_ll, P(c_ll)]),
    'effdet_weighted_median': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'effdet_sqnorm_workspace_floats': (c_ll, []),
    'effdet_sqnorm': (c_int, [c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_int]),
    'effdet_adam_clip_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_float, c_float, c_float,
                                      c_int, c_float, c_void_p]),
    'effdet_adam_clip_step_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_float, c_float, c_float, c_float,
                                          c_void_p, c_float, c_void_p]),
    'effdet_novelty_score': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int,
                                     c_void_p, c_void_p, c_void_p]),
    'effdet_ood_image_score': (c_int, [c_void_p, c_void_p, c_int, c_ll, c_void_p]),
    'effdet_auroc_counts': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'effdet_sepconv_meta': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_sepconv_tiles': (c_int, [c_int, c_int, c_void_p, c_void_p]),
    'effdet_bn_batch_stats': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float,
                                      c_void_p, c_void_p]),
    'effdet_topk_workspace_bytes': (c_ll, [c_int, c_ll]),
    'effdet_topk_select': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_ll, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_ll]),
    'effdet_decode_threshold_gather': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_decode_threshold': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    'effdet_nms_hard': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_detections_hard': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    'effdet_nms_soft': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_nms_soft_large': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_detection_loss_workspace_floats': (c_ll, [c_int, c_ll, c_int]),
    'effdet_detection_loss': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_ll, c_int,
                                      c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_ll]),
    'effdet_label_anchors_workspace_bytes': (c_ll, [c_int, c_int, c_ll]),
    'effdet_label_anchors': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_ll]),
    'effdet_train_gemm_nt': (c_int, [c_void_p, c_void_p, c_ll, c_ll, c_ll, c_void_p, c_void_p, c_void_p, c_ll, c_ll, c_ll,
                                     c_ll, c_int, c_int, c_int, c_void_p]),
    'effdet_train_gemm_tn_workspace_floats': (c_ll, [c_ll, c_int, c_int]),
    'effdet_train_gemm_tn': (c_int, [c_void_p, c_void_p, c_ll, c_ll, c_ll, c_void_p, c_ll, c_ll, c_ll, c_ll, c_int, c_int,
                                     c_void_p, c_void_p, c_ll]),
    'effdet_train_reduce_mid': (c_int, [c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p, c_int]),
    'effdet_train_dwconv_bwd_dx': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_train_dwconv_bwd_dw_workspace_floats': (c_ll, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_train_dwconv_bwd_dw': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p, c_ll, c_int]),
    'effdet_train_ew': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_float, c_float, c_float, c_float, c_ll, c_int, c_ll, c_void_p, c_void_p]),
    'effdet_train_col_reduce_workspace_floats': (c_ll, [c_int, c_ll, c_int]),
    'effdet_train_col_reduce': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_ll, c_int, c_void_p, c_void_p, c_ll, c_float]),
    'effdet_train_spatial': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    'effdet_train_im2col_stem': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    'effdet_train_se_bwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int]),
    'effdet_eval_match': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_eval_ap_workspace_bytes': (c_ll, [c_int]),
    'effdet_eval_ap': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_ll]),
    'effdet_train_fold_bn': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_train_convbn_grads': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    'effdet_train_bn_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'effdet_train_bn_bwd_prep': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_train_gemm_nt_levels': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, P(c_int), P(c_int),
                                            c_ll, c_ll, c_int, c_int, c_void_p]),
    'effdet_train_gemm_tn_levels': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, P(c_int), P(c_int), c_ll, c_ll, c_int, c_int,
                                            c_void_p, c_void_p, c_ll]),
    'effdet_train_levels_workspace_floats': (c_ll, [c_int, c_int, P(c_int), P(c_int), c_int]),
    'effdet_train_levels_dw': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, P(c_int), P(c_int), c_int, c_int]),
    'effdet_train_levels_dw_bwd_dw': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, P(c_int), P(c_int), c_int,
                                              c_void_p, c_ll, c_int]),
    'effdet_train_levels_col_reduce': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, P(c_float), c_int, c_int,
                                               P(c_int), P(c_int), c_int, c_void_p, c_void_p, c_ll]),
    'effdet_train_levels_bn_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, P(c_void_p), P(c_void_p), P(c_void_p),
                                                P(c_void_p), P(c_void_p), P(c_int), P(c_float), P(c_float), P(c_float), P(c_float),
                                                c_void_p, c_void_p, c_void_p, c_void_p]),
    'effdet_train_levels_bn_bwd_prep': (c_int, [c_void_p, c_void_p, c_void_p, P(c_float), c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    'effdet_train_levels_ew': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, P(c_int), c_int, c_int, P(c_int), P(c_int), c_int]),
    'effdet_train_fpn_weights': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'effdet_train_fpn_combine': (c_int, [c_void_p, c_int, P(c_void_p), P(c_int), P(c_int), c_int, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_int]),
    'effdet_train_fpn_dots_workspace_floats': (c_ll, [c_int, c_int, c_int, c_int]),
    'effdet_train_fpn_wgrad': (c_int, [c_void_p, c_int, P(c_void_p), P(c_int), P(c_int), c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_ll]),
    'effdet_train_fpn_input_bwd': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int]),
    'effdet_train_dwconv_fwd_parts': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'effdet_train_dwconv_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_train_se_gate': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int]),
    'effdet_train_gemm_nt_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_ll, c_int, c_int]),
    'effdet_train_gemm_tn_scaled': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_ll, c_int, c_int, c_void_p, c_void_p, c_ll]),
    'effdet_train_dwconv_bwd_dx_silu': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    'effdet_train_prep_table': (c_int, [c_void_p, c_void_p, c_int, c_ll]),
    'effdet_train_grads_table': (c_int, [c_void_p, c_void_p, c_int, c_int]),
    'effdet_train_bn_var_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_ll]),
    'effdet_train_bn_bwd_sums': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_ll]),
    'effdet_gather_ood': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p]),
}


def build(verbose=False):
    """Compile every HIP source for gfx950 into libeffdet_hip.so (in-tree)."""
    jobs = str(min(8, os.cpu_count() or 1))
    r = subprocess.run(['make', '-C', CSRC, '-j', jobs], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError('building libeffdet_hip.so failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB_PATH


TEST_VARIANTS = {
    # tag -> (translation unit, defines): libeffdet_hip_<tag>.so beside the product library.  Loaded by ONE test each, never by the
    # package.  spin0: mbconv_wide.hip whose waves give up their hand-off poll at once, so that the device-side failure report
    # (effdet_device_error) can be seen once (tests/test_kernels_gpu.py::test_wide_handoff_timeout_is_reported).
    'spin0': ('mbconv_wide', '-DWIDE_SPIN_LIMIT=0'),
}


def build_test_variants(verbose=False):
    paths = []
    for tag, (unit, defs) in TEST_VARIANTS.items():
        r = subprocess.run(['make', '-C', CSRC, 'variant', 'TAG=' + tag, 'UNIT=' + unit, 'VDEFS=' + defs], capture_output=not verbose, text=True)
        if r.returncode != 0:
            raise RuntimeError('building the %s variant failed:\n%s\n%s' % (tag, r.stdout, r.stderr))
        paths.append(os.path.join(_HERE, 'libeffdet_hip_%s.so' % tag))
    return paths


def load():
    """Load the shared library; RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # The library must share ONE HIP runtime with the host framework: torch bundles its own
    # libamdhip64/libhsa-runtime64, and whichever copy is mapped first wins the SONAME.  Import torch
    # first so that its runtime is the one both sides use (loading this library first maps /opt/rocm's
    # copy and the process then sees "no ROCm-capable device").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('%s is missing - run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(there is no CPU fallback for the HIP path)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.effdet_abi_version() != 1:
        raise RuntimeError('libeffdet_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        detail = ''
        if rc == -5 and _lib is not None:
            detail = ' (%s)' % (_lib.effdet_last_error() or b'').decode()
        raise RuntimeError('%s failed with code %d%s' % (what, rc, detail))
