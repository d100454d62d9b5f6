"""ctypes binding of the C ABI in include/pgt_hip.h.

The product path loads exactly one library: pytorch_geometric_temporal_amd/lib/libpgt_hip.so (gfx950 code objects).
There is NO CPU fallback: if the library is missing, or a tensor is not on a HIP device, the ops raise.
(The test-suite can inject a *test double* built from the same kernel sources — see `_set_library_for_testing` —
which only accepts CPU tensors and announces itself as target "emu".)
"""
import ctypes
import os
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpgt_hip.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_ptr = ctypes.c_void_p
c_size = ctypes.c_size_t
Ptr3 = c_ptr * 3                      # a host array of three device pointers (the z, r, h parameters of a gated cell)
c_p3 = ctypes.POINTER(Ptr3)


class PgtError(RuntimeError):
    pass


class PgtLibraryMissing(PgtError):
    pass


class CsrStruct(ctypes.Structure):
    _fields_ = [("rowptr", c_ptr), ("col", c_ptr), ("val", c_ptr)]


class EllwStruct(ctypes.Structure):
    _fields_ = [("slots", c_ptr), ("vals", c_ptr), ("scale", c_ptr), ("tile_rows", ctypes.c_int32),
                ("halo", ctypes.c_int32), ("width", ctypes.c_int32), ("config", ctypes.c_int32), ("n_tiles", c_i64),
                ("far_col", c_ptr), ("far_rows", ctypes.c_int32), ("order", c_ptr),
                ("hub_col", c_ptr), ("hub_val", c_ptr), ("hub_rows", c_ptr), ("hub_partial", c_ptr),
                ("n_hub", ctypes.c_int32), ("hub_split", ctypes.c_int32), ("far_src", c_ptr)]


class DConvGraphStruct(ctypes.Structure):
    _fields_ = [("fwd_o", CsrStruct), ("fwd_i", CsrStruct), ("bwd_o", CsrStruct), ("bwd_i", CsrStruct),
                ("deg_out", c_ptr), ("deg_in", c_ptr), ("info", c_ptr)]


class RowMapStruct(ctypes.Structure):
    """pgt_rowmap: row m of an operand at base + (m // period) * stride_hi + (m % period) * ld."""
    _fields_ = [("period", c_i64), ("stride_hi", c_i64)]


class SymGraphStruct(ctypes.Structure):
    _fields_ = [("fwd", CsrStruct), ("bwd", CsrStruct), ("deg", c_ptr), ("info", c_ptr)]


# name -> (restype, argtypes); mirrors include/pgt_hip.h one to one (tests/test_cabi.py checks the header against this)
PROTOTYPES = {
    "pgt_abi_version": (c_int, []),
    "pgt_last_error": (ctypes.c_char_p, []),
    "pgt_build_target": (ctypes.c_char_p, []),
    "pgt_tune": (c_int, [ctypes.c_char_p, c_int]),
    "pgt_prep_workspace_bytes": (c_size, [c_i64, c_i64]),
    "pgt_dconv_prep": (c_int, [c_ptr, c_ptr, c_i64, c_i64, ctypes.POINTER(DConvGraphStruct), c_ptr, c_size, c_ptr]),
    "pgt_gcn_prep": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, ctypes.POINTER(SymGraphStruct), c_ptr,
                             c_size, c_ptr]),
    "pgt_cheb_prep": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_f32, c_int, ctypes.POINTER(SymGraphStruct),
                              c_ptr, c_size, c_ptr]),
    "pgt_cheb_prep_graphs": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_i64, c_int,
                                     ctypes.POINTER(SymGraphStruct), c_ptr, c_size, c_ptr]),
    "pgt_spmm_csr_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_f32,
                                 c_f32, c_i64, c_ptr]),
    "pgt_ellw_plan": (c_int, [c_i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                              ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_i64),
                              ctypes.POINTER(ctypes.c_int32)]),
    "pgt_ellw_build": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, ctypes.POINTER(EllwStruct), c_ptr, c_ptr, c_ptr, c_ptr,
                               c_ptr, c_ptr, c_ptr]),
    "pgt_spmm_ellw_f32": (c_int, [ctypes.POINTER(EllwStruct), c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                  c_ptr, c_i64, c_f32, c_f32, c_i64, c_ptr]),
    "pgt_tile_order_host": (c_int, [c_ptr, c_ptr, c_i64, ctypes.c_int32, ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "pgt_csr_locality": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, ctypes.c_int32, c_ptr]),
    "pgt_spmm_csr_long_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, ctypes.c_int32, c_ptr, c_i64, c_ptr, c_i64,
                                      c_ptr, c_i64, c_f32, c_f32, c_i64, c_ptr]),
    "pgt_spmm_csr_rows_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                      c_ptr, c_i64, c_f32, c_f32, c_i64, c_ptr]),
    "pgt_spmm_csr_att_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_int, c_ptr]),
    "pgt_sddmm_att_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "pgt_dconv_stack_slab_fits": (c_int, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    "pgt_dconv_stack_slab_plan": (c_int, [c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_dconv_stack_slab_f32": (c_int, [ctypes.POINTER(CsrStruct), ctypes.POINTER(CsrStruct), c_i64, c_i64, c_i64,
                                         c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr]),
    "pgt_dconv_stack_slab_bwd_f32": (c_int, [ctypes.POINTER(CsrStruct), ctypes.POINTER(CsrStruct), c_i64, c_i64, c_i64,
                                             c_i64, c_i64, c_i64, c_ptr, c_i64, c_int, c_ptr]),
    "pgt_gemm_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_i64,
                             c_ptr, c_i64, c_i64, c_int, c_ptr]),
    "pgt_gemm_gru_zr_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64,
                                    c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_gemm_gru_h_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                                   c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_gemm_tn_acc_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                    c_i64, c_ptr]),
    "pgt_gemm_tn_det_ws_bytes": (c_size, [c_i64, c_i64, c_i64, c_i64]),
    "pgt_gemm_tn_det_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64,
                                    c_i64, c_ptr, c_size, c_ptr]),
    "pgt_gru_zr_f32": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_gru_h_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_gru_h_bwd_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                                  c_ptr, c_ptr, c_i64, c_int, c_i64, c_i64, c_ptr]),
    "pgt_gru_zr_bwd_f32": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64,
                                   c_ptr]),
    "pgt_lstm_gates_f32": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_lstm_gates_bwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64,
                                       c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr]),
    "pgt_copy2d_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_add2d_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_axpby2d_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_f32, c_ptr, c_i64, c_f32, c_i64, c_i64, c_ptr]),
    "pgt_swap01_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "pgt_att_sigmoid_scores_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    "pgt_att_softmax_rows_f32": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "pgt_att_softmax_rows_bwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "pgt_att_sigmoid_bwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "pgt_window_gather_f32": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_int, c_ptr]),
    "pgt_adam_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_ptr]),
    "pgt_relu_linear_fits": (c_int, [c_i64, c_i64]),
    "pgt_relu_linear_bwd_ws_floats": (c_i64, [c_i64, c_i64]),
    "pgt_relu_linear_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "pgt_relu_linear_bwd_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr,
                                        c_ptr, c_i64, c_ptr]),
    "pgt_dcrnn_pack_weights_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "pgt_dcrnn_unpack_weight_grads_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                                  c_ptr]),
    "pgt_dcrnn_stage_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "pgt_dcrnn_cell_k1_fits": (c_int, [c_i64, c_i64, c_i64]),
    "pgt_dcrnn_cell_k1_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                                      c_i64, c_i64, c_i64, c_ptr]),
    "pgt_dcrnn_cell_k1_bwd_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                          c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64,
                                          c_ptr]),
    "pgt_gcn_small_fits": (c_int, [c_i64, c_i64, c_i64, c_i64]),
    "pgt_gcn_small_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr,
                                  c_ptr, c_ptr]),
    "pgt_gcn_small_bwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64,
                                      c_ptr, c_ptr, c_i64, c_ptr]),
    "pgt_evolve_weight_f32": (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr,
                                      c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "pgt_evolve_weight_bwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                          c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "pgt_bmm_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64,
                            c_i64, c_i64, c_int, c_ptr]),
    "pgt_relu_layernorm_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "pgt_relu_layernorm_bwd_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr,
                                           c_ptr]),
    "pgt_tconv_glu_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                  c_ptr]),
    "pgt_tconv_glu_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    "pgt_batchnorm_nodes_f32": (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_int, c_ptr,
                                        c_ptr, c_ptr]),
    "pgt_batchnorm_nodes_bwd_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr,
                                            c_ptr]),
    "pgt_tgcn_pack_weights_f32": (c_int, [c_p3, c_p3, c_p3, c_p3, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "pgt_tgcn_unpack_weight_grads_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_p3, c_p3, c_p3, c_i64, c_i64, c_p3, c_p3, c_p3,
                                                 c_p3, c_ptr]),
    "pgt_dcrnn_seq_small_fits": (c_int, [c_i64, c_i64, c_i64, c_i64, c_i64, c_i64]),
    "pgt_dcrnn_seq_small_save_floats": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "pgt_dcrnn_seq_small_f32": (c_int, [ctypes.POINTER(CsrStruct), ctypes.POINTER(CsrStruct), c_i64, c_i64, c_i64, c_ptr, c_i64,
                                        c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64,
                                        c_i64, c_ptr, c_ptr]),
    "pgt_dcrnn_seq_small_bwd_f32": (c_int, [ctypes.POINTER(CsrStruct), ctypes.POINTER(CsrStruct), c_i64, c_i64, c_i64, c_ptr, c_i64,
                                            c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64,
                                            c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "pgt_dcrnn_seq64_fits": (c_int, [c_i64, c_i64, c_i64, c_i64, c_i64, c_i64]),
    "pgt_dcrnn_seq64_pack_floats": (c_i64, [c_i64]),
    "pgt_dcrnn_seq64_pack_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "pgt_dcrnn_seq64_f32": (c_int, [ctypes.POINTER(CsrStruct), ctypes.POINTER(CsrStruct), c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64,
                                    c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64,
                                    c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "pgt_dcrnn_seq64_pack_bwd_f32": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "pgt_dcrnn_seq64_bwd_ws_floats": (c_i64, [c_i64, c_i64]),
    "pgt_dcrnn_seq64_bwd_f32": (c_int, [ctypes.POINTER(CsrStruct), ctypes.POINTER(CsrStruct), c_i64, c_i64, c_i64, c_ptr, c_i64, c_i64,
                                        c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr,
                                        c_ptr, c_i64, c_ptr]),
    "pgt_tgcn_cell_fits": (c_int, [c_i64, c_i64]),
    "pgt_tgcn_cell_bwd_ws_floats": (c_i64, [c_i64, c_i64]),
    "pgt_tgcn_cell_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr,
                                  c_i64, c_ptr]),
    "pgt_tgcn_cell_bwd_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64,
                                      c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "pgt_tgcn_cell_bwd_acc_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64,
                                          c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
}

EXPECTED_ABI = 17


class PgtLib:
    """A loaded libpgt_*.so with typed entry points.  `target` is "gfx950" (product) or "emu" (test double)."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise PgtLibraryMissing(
                f"{path} not found. Build it with `python -m pytorch_geometric_temporal_amd._build` "
                f"(or __graft_entry__.build()). There is no CPU fallback for the HIP kernels.")
        self.path = path
        self._dll = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise PgtError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)
        abi = self._pgt_abi_version()
        if abi != EXPECTED_ABI:
            raise PgtError(f"{path}: ABI version {abi}, expected {EXPECTED_ABI}")
        self.target = self._pgt_build_target().decode()

    def last_error(self):
        return self._pgt_last_error().decode()

    def call(self, name, *args):
        rc = getattr(self, "_" + name)(*args)
        if rc != 0:
            raise PgtError(f"{name} failed with code {rc}: {self.last_error()}")

    def tune(self, key, value):
        self.call("pgt_tune", key.encode(), int(value))

    def prep_workspace_bytes(self, E, N):
        return int(self._pgt_prep_workspace_bytes(E, N))


_LIB = None
_TESTING = False


def _preload_hip_runtime():
    # libpgt_hip.so needs libamdhip64.so.7; make sure the copy PyTorch already mapped is the one it binds to, so
    # device pointers and hipStream_t handles created by torch are valid inside the library.
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(torch_lib):
        try:
            ctypes.CDLL(torch_lib, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def _check_not_stale():
    """Refuse a library that was built from other kernel sources than the ones lying next to it (only checked when the
    sources are present, i.e. in a source checkout): a stale .so would silently run - and benchmark - old kernels."""
    from . import _build
    if not (os.path.exists(LIB_PATH) and os.path.exists(_build.BUILD_ID_PATH) and os.path.isdir(_build.CSRC)):
        return
    try:
        want = _build.source_digest()
    except OSError:
        return
    with open(_build.BUILD_ID_PATH) as fh:
        have = fh.read().strip()
    if have != want and os.environ.get("PGT_ALLOW_STALE_LIB") != "1":
        raise PgtError(f"{LIB_PATH} was built from different sources than pytorch_geometric_temporal_amd/csrc "
                       f"(build id {have[:12]}, sources {want[:12]}): rebuild with "
                       "`python -m pytorch_geometric_temporal_amd._build`")


def get_lib():
    """The product library (gfx950).  Raises PgtLibraryMissing when it has not been built."""
    global _LIB
    if _LIB is None:
        _preload_hip_runtime()
        _check_not_stale()
        _LIB = PgtLib(LIB_PATH)
        if _LIB.target != "gfx950":
            raise PgtError(f"{LIB_PATH} reports target {_LIB.target!r}, expected 'gfx950'")
        # A/B switches for benchmarking without code changes: PGT_TUNE="gemm_dbp=1,slab_pairs=1" (keys of pgt_tune)
        for item in filter(None, os.environ.get("PGT_TUNE", "").split(",")):
            key, _, val = item.partition("=")
            _LIB.tune(key.strip(), int(val))
    return _LIB


def _set_library_for_testing(lib):
    """TEST-ONLY hook: route the host logic to a test double (tests/_emu/libpgt_emu.so).  Never used by the product."""
    global _LIB, _TESTING
    if lib is not None:
        if lib.target != "emu":
            raise PgtError("only the 'emu' test double may be injected")
        warnings.warn("pytorch_geometric_temporal_amd: running on the CPU TEST DOUBLE of the HIP kernels", stacklevel=2)
    _LIB = lib
    _TESTING = lib is not None


def check_tensor(lib, t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if lib.target == "gfx950":
        if not t.is_cuda:
            raise PgtError(f"{name} is on {t.device}; the HIP kernels need a GPU tensor (no CPU fallback)")
    else:
        if t.is_cuda:
            raise PgtError(f"{name}: the emu test double only takes CPU tensors")
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(lib, t):
    """torch's current stream on t's device as the ABI's pgt_stream_t (plain int: ctypes converts it; no argument
    objects per call — a step of the launch-bound configs makes hundreds of these)."""
    if lib.target == "gfx950":
        if _raw_stream is not None:
            return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def ptr3(a, b, c):
    """Three device pointers as the host array a `const float* const [3]` parameter takes (None = NULL)."""
    return ctypes.byref(Ptr3(ptr(a), ptr(b), ptr(c)))


def ptr(t):
    """Device address as a plain int (None = NULL): ctypes converts it per its argtypes / structure field."""
    return t.data_ptr() if t is not None else None
