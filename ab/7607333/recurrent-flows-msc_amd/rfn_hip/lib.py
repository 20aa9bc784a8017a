"""ctypes binding of librfn_hip.so (the C ABI declared in include/rfn_hip.h).

The library is built in-tree by `make -C recurrent-flows-msc_amd/csrc` (see __graft_entry__.build) and is the
ONLY compute backend of this package: if it cannot be loaded, or a tensor is not a contiguous fp32 device tensor,
the call raises — there is no CPU / eager fallback.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librfn_hip.so")

_c_f = ctypes.c_void_p  # device float*
_c_i = ctypes.c_int
_c_l = ctypes.c_long
_c_s = ctypes.c_void_p  # hipStream_t

# name -> argtypes (restype is int unless listed in _RESTYPES); order mirrors include/rfn_hip.h
SIGNATURES = {
    "rfn_abi_version": [],
    "rfn_last_error": [],
    "rfn_squeeze2d_f32": [_c_f, _c_l, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_channel_stats_f32": [_c_f, _c_l, _c_f, _c_f, _c_i, _c_i, _c_i, _c_s],
    "rfn_actnorm_invconv_fwd_f32": [_c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_i, _c_i, _c_i, _c_s],
    "rfn_actnorm_invconv_bwd_f32": [_c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f,
                                    _c_i, _c_i, _c_i, _c_s],
    "rfn_invconv_weights_fwd_f32": [ctypes.c_void_p] * 5 + [_c_f, _c_f, _c_i, _c_i, _c_i, _c_s],
    "rfn_invconv_weights_bwd_f32": [ctypes.c_void_p] * 5 + [_c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_s],
    "rfn_invconv_actnorm_rev_f32": [_c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_i, _c_i, _c_i, _c_s],
    "rfn_conv2d_fwd_f32": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i,
                           _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_i, _c_s],
    "rfn_conv2d_fwd_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i,
                              _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_i, _c_s],
    "rfn_packed_weight_size_bf16x3": [_c_i, _c_i, _c_i],
    "rfn_pack_conv_weight_bf16x3": [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_packed_weight_size_bf16x6": [_c_i, _c_i, _c_i],
    "rfn_pack_conv_weight_bf16x6": [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_conv2d_fwd_bf16x6": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i,
                              _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_i, _c_s],
    "rfn_conv2d_dgrad_act_rows_bf16x3": [_c_i, _c_i, _c_i, _c_i, _c_i, _c_i],
    "rfn_conv2d_dgrad_act_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_f, _c_i, _c_f, _c_l, _c_f, _c_i, _c_i, _c_i,
                                    _c_i, _c_i, _c_s],
    "rfn_pack_conv_weights_batched_bf16x3": [_c_f, _c_i, _c_s],
    "rfn_gather_affine_f32": [_c_f, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i,
                              _c_i, _c_s],
    "rfn_affine_zeros_bwd_f32": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l, _c_f,
                                 _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_glow_shell_fwd_f32": [_c_f, _c_l, _c_f, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_f, _c_f,
                               _c_f, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_glow_shell_bwd_f32": [_c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_f, _c_f,
                               _c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_actnorm_invconv_bwd_ld_f32": [_c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_i,
                                       _c_i, _c_i, _c_s],
    "rfn_dgrad_small_supported": [_c_i, _c_i, _c_i, _c_i, _c_i],
    "rfn_conv3x3_smallcout_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                     _c_i, _c_s],
    "rfn_gemm_wgrad_grouped_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_i, _c_i, _c_i, _c_s],
    "rfn_conv3x3_wgrad_implicit_grouped_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_i, _c_i,
                                                  _c_i, _c_i, _c_s],
    "rfn_coupling_po_supported": [_c_i, _c_i, _c_i, _c_i, _c_i, _c_i],
    "rfn_coupling_po_packed_bytes": [_c_i, _c_i],
    "rfn_coupling_po_pack": [_c_f, _c_i, _c_s],
    "rfn_coupling_po_fwd": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_l,
                            _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_glow_shell_fwd_ld_floats": [_c_i, _c_i, _c_i, _c_i],
    "rfn_logdet_reduce_f32": [_c_f, _c_i, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_coupling_po_mask_floats": [_c_i, _c_i, _c_i],
    "rfn_coupling_po_bwd_supported": [_c_i, _c_i, _c_i, _c_i],
    "rfn_coupling_po_bwd_packed_bytes": [_c_i],
    "rfn_coupling_po_pack_bwd": [_c_f, _c_i, _c_s],
    "rfn_coupling_po_bwd_part_floats": [_c_i, _c_i, _c_i],
    "rfn_coupling_po_bwd": [_c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_i, _c_i, _c_i, _c_i,
                            _c_i, _c_s],
    "rfn_coupling_po_bwd_finish": [ctypes.c_void_p] * 8 + [_c_i, _c_i, _c_i, _c_s],
    "rfn_gemm_wgrad_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_i, _c_i, _c_s],
    "rfn_conv3x3_wgrad_implicit_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_i, _c_i, _c_i,
                                          _c_s],
    "rfn_im2col3x3_f32": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_i, _c_i, _c_i, _c_s],
    "rfn_packed_weight_size": [_c_i, _c_i, _c_i],
    "rfn_pack_conv_weight_f32": [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_conv2d_wgrad_f32": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_wgrad_finish_f32": [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_tap_gather_f32": [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_tap_scatter_f32": [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_conv_epilogue_bwd_f32": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i,
                                  _c_s],
    "rfn_affine_coupling_f32": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_affine_coupling_bwd_f32": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l,
                                    _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_gauss_logp_f32": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_gauss_logp_bwd_f32": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_l, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i,
                               _c_s],
    "rfn_gauss_sample_f32": [_c_f, _c_l, _c_f, _c_f, _c_l, ctypes.c_float, _c_i, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_latent_step_fwd_f32": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_s],
    "rfn_latent_step_bwd_f32": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i,
                                _c_s],
    "rfn_conv3x3_fewcin_supported": [_c_i, _c_i],
    "rfn_conv3x3_fewcin_fwd_f32": [_c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_conv3x3_c1_wgrad16_f32": [_c_f, _c_l, _c_f, _c_l, _c_f, _c_i, _c_i, _c_i, _c_s],
    "rfn_smallmap_packed_size": [_c_i, _c_i, _c_i, _c_i, _c_i],
    "rfn_smallmap_pack_bf16x3": [_c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_s],
    "rfn_smallmap_pack_batched_bf16x3": [ctypes.c_void_p, _c_i, _c_s],
    "rfn_pack_conv_weights_hostdescs_bf16x3": [ctypes.c_void_p, _c_i, _c_s],
    "rfn_smallmap_dense_bf16x3": [_c_f, _c_f, ctypes.c_float, _c_f, _c_f, _c_f, _c_i, ctypes.c_float, _c_f, _c_f, _c_i,
                                  _c_i, _c_i, _c_i, _c_s],
    "rfn_smallmap_dense_pair_bf16x3": [_c_f, _c_f, ctypes.c_float, _c_f, _c_f, _c_f, _c_i, ctypes.c_float, _c_f, _c_f, _c_i,
                                       _c_i, _c_f, _c_f, ctypes.c_float, _c_f, _c_f, _c_f, _c_i, ctypes.c_float, _c_f, _c_f,
                                       _c_i, _c_i, _c_i, _c_i, _c_s],
    "rfn_smallmap_conv_bf16x3": [_c_f, _c_l, _c_i, _c_f, _c_l, _c_i, _c_f, _c_f, _c_l, _c_f, _c_l, _c_i, _c_i, _c_i, _c_i,
                                 _c_i, _c_i, _c_i, _c_f, _c_f, _c_i, _c_s],
    "rfn_stepbn_scratch_floats": [_c_i, _c_i, _c_i],
    "rfn_stepbn_fwd_f32": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, ctypes.c_float, ctypes.c_void_p,
                           _c_i, _c_i, _c_i, _c_i, ctypes.c_float, _c_i, ctypes.c_float, _c_s],
    "rfn_stepbn_bwd_f32": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, ctypes.c_float,
                           _c_i, ctypes.c_float, _c_i, _c_i, _c_s],
    "rfn_stepbn_apply_f32": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, ctypes.c_float, _c_i, ctypes.c_float,
                             _c_s],
    "rfn_adam_chunk_elems": [],
    "rfn_adam_step_f32": [ctypes.c_void_p, ctypes.c_void_p, _c_i, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                          ctypes.c_double, ctypes.c_double, _c_i, _c_s],
    "rfn_convlstm_gates_fwd_f32": [_c_f, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_i, _c_i, _c_i,
                                   _c_s],
    "rfn_convlstm_gates_bwd_f32": [_c_f, _c_f, _c_l, _c_f, _c_l, _c_f, _c_l, _c_f, _c_l, _c_f, _c_f, _c_f, _c_f, _c_f,
                                   _c_l, _c_i, _c_i, _c_i, _c_s],
}
_RESTYPES = {"rfn_last_error": ctypes.c_char_p, "rfn_stepbn_scratch_floats": ctypes.c_long, "rfn_packed_weight_size": ctypes.c_long,
             "rfn_packed_weight_size_bf16x3": ctypes.c_long, "rfn_packed_weight_size_bf16x6": ctypes.c_long,
             "rfn_smallmap_packed_size": ctypes.c_long,
             "rfn_coupling_po_packed_bytes": ctypes.c_long, "rfn_coupling_po_mask_floats": ctypes.c_long, "rfn_glow_shell_fwd_ld_floats": ctypes.c_long,
             "rfn_coupling_po_bwd_packed_bytes": ctypes.c_long, "rfn_coupling_po_bwd_part_floats": ctypes.c_long}

_lib = None


def load():
    """Load librfn_hip.so (once).  Raises RuntimeError if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "librfn_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C recurrent-flows-msc_amd/csrc`. There is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


# Optional in-process kernel timing (bench.py): when PROFILE is a list, every call is bracketed by HIP events recorded
# on the stream the kernel is launched on, and (name, meta, start_event, end_event) is appended.
PROFILE = None


# work queued by rfn_hip.ops that must be launched before the next kernel (weight packs collected into one launch): a
# callable, run -- once -- at the top of the next call()
PENDING_FLUSH = None


def call(name, *args, meta=None):
    """Invoke an int-returning entry point on the current torch stream; raise on a non-zero code."""
    global PENDING_FLUSH
    if PENDING_FLUSH is not None:
        flush, PENDING_FLUSH = PENDING_FLUSH, None
        flush()
    lib = load()
    cur = torch.cuda.current_stream()
    stream = ctypes.c_void_p(cur.cuda_stream)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        rc = getattr(lib, name)(*args, stream)
        e1.record(cur)
        PROFILE.append((name, meta, e0, e1))
    else:
        rc = getattr(lib, name)(*args, stream)
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (name, rc, lib.rfn_last_error().decode()))


def ptr_array(tensors, name="tensor"):
    """host array of device pointers (for the grouped entry points); the caller keeps the tensors alive"""
    arr = (ctypes.c_void_p * len(tensors))(*[dev(t, name, check_contiguous=False).value for t in tensors])
    return arr


def dev(t, name="tensor", check_contiguous=True):
    """Validate a device tensor for the kernels and return its pointer: fp32, on the GPU, contiguous (frames() checks
    the per-frame layout of channel-slice views itself and passes check_contiguous=False)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("rfn_hip kernels need device tensors; %s is on %s (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("rfn_hip kernels are fp32; %s is %s" % (name, t.dtype))
    if check_contiguous and not t.is_contiguous():
        raise RuntimeError("rfn_hip kernels need dense tensors; %s has shape %s stride %s" %
                           (name, tuple(t.shape), tuple(t.stride())))
    return ctypes.c_void_p(t.data_ptr())


def frames(t, name="tensor"):
    """(pointer, frame stride) of an [N,C,H,W] (or [N,C,HW]) tensor whose per-frame block is dense."""
    p = dev(t, name, check_contiguous=False)
    shape, stride = t.shape, t.stride()
    exp = 1
    for d in range(t.dim() - 1, 0, -1):
        if shape[d] != 1 and stride[d] != exp:
            raise RuntimeError("%s: per-frame layout must be dense NCHW, got shape %s stride %s" %
                               (name, tuple(shape), tuple(stride)))
        exp *= shape[d]
    ns = stride[0] if shape[0] > 1 else exp
    return p, int(ns)
