"""ctypes binding of libaaerec_hip.so (C ABI: include/aaerec_hip.h).

PyTorch is used for plumbing only: it owns the HBM arena (one uint8 tensor), gives the
current HIP stream and lets parameter / gradient views be handed to torch.distributed
(RCCL).  All arithmetic of the AAE step runs in the hand-written gfx950 kernels.

There is NO CPU fallback: importing this module without the built library, or creating a
model without a GPU, raises.
"""
import ctypes as C
import itertools
import weakref
import os

import numpy as np
import torch  # must be imported before the library so both share one HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AAE_HIP_LIB") or os.path.join(_HERE, "libaaerec_hip.so")     # (AAE_HIP_LIB: A/B builds of the library)

ABI_VERSION = 4
# getattr(nn, activation)() of the reference (aae.py:110): the parameter-free element-wise classes of torch.nn at their default
# arguments (r6: 6-19).  Classes with parameters, state or a row-wise definition (PReLU, RReLU, Threshold, GLU, Softmax,
# Softmin, LogSoftmax, MultiheadAttention ...) and Tanhshrink have no kernel: HipAAE raises with this list.
ACTIVATIONS = {"ReLU": 0, "SELU": 1, "Tanh": 2, "Sigmoid": 3, "ELU": 4, "LeakyReLU": 5,
               "Softplus": 6, "Hardtanh": 7, "ReLU6": 8, "CELU": 9, "Softsign": 10, "Hardsigmoid": 11, "LogSigmoid": 12,
               "Softshrink": 13, "Hardshrink": 14, "Identity": 15, "GELU": 16, "SiLU": 17, "Mish": 18, "Hardswish": 19}
FINALS = {"linear": 0, "softmax": 1, "sigmoid": 2}
OPTIMIZERS = {"adam": 0, "sgd": 1}
PRIORS = {"gauss": 0, "categorical": 1, "bernoulli": 2}
RNG_INJECT, RNG_DEVICE = 0, 1
GRAD_FUSED, GRAD_EXPORT = 0, 1


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()

# tensor ids (aaerec_hip.h)
T_ENC_W1T, T_ENC_B1, T_ENC_W2, T_ENC_W3, T_DEC_V1, T_DEC_V2, T_DEC_V3, T_DISC_D1, T_DISC_D2, T_DISC_D3 = range(10)
T_ADAM_ENC, T_ADAM_GEN, T_ADAM_DEC, T_ADAM_DISC, T_GRAD = 16, 32, 48, 64, 80
T_ACT_Z, T_ACT_LOSSES, T_ACT_A1, T_ACT_DZC, T_ACT_DH2, T_ACT_DA2 = 96, 97, 98, 99, 100, 101
T_ACT_GA1 = 102
CAT_SUM, CAT_MEAN = 0, 1
CAT_SPARSE_ADAM, CAT_ADAM = 0, 1
O_ENC, O_DEC, O_GEN, O_DISC = 0, 1, 2, 3
MODEL_AAE, MODEL_AE, MODEL_VAE = 0, 1, 3          # cfg.model_kind
DTYPE_F32, DTYPE_BF16 = 0, 1                      # cfg.dtype


class AaeConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_items", C.c_int32), ("n_hidden", C.c_int32),
                ("n_code", C.c_int32), ("cond_inc", C.c_int32), ("max_batch", C.c_int32),
                ("max_nnz", C.c_int32), ("activation", C.c_int32), ("enc_final", C.c_int32),
                ("optimizer", C.c_int32), ("normalize_inputs", C.c_int32), ("rng_mode", C.c_int32),
                ("prior", C.c_int32), ("grad_mode", C.c_int32), ("dropout1", C.c_float),
                ("dropout2", C.c_float), ("gen_lr", C.c_float), ("reg_lr", C.c_float),
                ("prior_scale", C.c_float), ("has_prior_scale", C.c_int32), ("seed", C.c_uint64),
                ("unfused_decoder", C.c_int32), ("dp_world", C.c_int32), ("model_kind", C.c_int32),
                ("dtype", C.c_int32), ("blocked_output", C.c_int32), ("dense_noise", C.c_int32),
                ("reserved", C.c_int32 * 2)]


class AaeBatch(C.Structure):
    _fields_ = [("indptr_dev", C.c_void_p), ("indices_dev", C.c_void_p), ("values_dev", C.c_void_p),
                ("rows_dev", C.c_void_p), ("row_start", C.c_int32), ("n_rows", C.c_int32),
                ("nnz_bound", C.c_int32), ("max_row_nnz", C.c_int32), ("generation", C.c_int64)]


class AaeRngInject(C.Structure):
    _fields_ = [("masks_dev", C.c_void_p * 12), ("z_real_dev", C.c_void_p)]


_COLL_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
_COLL_AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class AaeCollectives(C.Structure):
    """aae_collectives (include/aaerec_hip.h): the collectives aae_dp_step calls at its exchange points."""
    _fields_ = [("ctx", C.c_void_p), ("all_gather", _COLL_AG), ("reduce_scatter", _COLL_AG), ("all_reduce", _COLL_AR),
                ("world", C.c_int32), ("rank", C.c_int32)]


class AaeTensor(C.Structure):
    _fields_ = [("byte_offset", C.c_size_t), ("rows", C.c_int64), ("cols", C.c_int64), ("ld", C.c_int64)]


_PROTOS = {
    "aae_abi_version": (C.c_int, []),
    "aae_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "aae_last_error": (C.c_char_p, []),
    "aae_arena_bytes": (C.c_int, [C.POINTER(AaeConfig), C.POINTER(C.c_size_t)]),
    "aae_create": (C.c_int, [C.POINTER(AaeConfig), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "aae_destroy": (C.c_int, [C.c_void_p]),
    "aae_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(AaeTensor)]),
    "aae_set_lr": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "aae_params_changed": (C.c_int, [C.c_void_p]),
    "aae_set_rng_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "aae_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aae_load_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aae_store_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aae_load_adam": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "aae_store_adam": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "aae_step": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.POINTER(AaeRngInject), C.c_void_p]),
    "aae_ae_encode": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.POINTER(AaeRngInject), C.c_void_p, C.c_void_p]),
    "aae_ae_decode_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(AaeRngInject), C.c_void_p, C.c_void_p]),
    "aae_vae_step": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_void_p, C.c_void_p]),
    "aae_vae_predict": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p]),
    "aae_vae_encode": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "aae_vae_decode_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "aae_vae_encoder_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "aae_decoder_step": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_int64, C.POINTER(AaeRngInject),
                                   C.c_void_p, C.c_void_p]),
    "aae_ae_encoder_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "aae_disc_gen": (C.c_int, [C.c_void_p, C.POINTER(AaeRngInject), C.c_void_p]),
    "aae_disc_step": (C.c_int, [C.c_void_p, C.POINTER(AaeRngInject), C.c_void_p]),
    "aae_gen_step": (C.c_int, [C.c_void_p, C.POINTER(AaeRngInject), C.c_void_p]),
    "aae_read_losses": (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 3), C.c_void_p]),
    "aae_predict": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "aae_predict_topk": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aae_rank_max_rows": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "aae_decode_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(AaeBatch), C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "aae_encode": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_void_p]),
    "aae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "aae_apply_updates": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "aae_apply_updates_except": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "aae_apply_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "aae_w1_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "aae_w1_packet_floats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "aae_w1_import": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int, C.c_void_p]),
    "aae_set_grad_scale": (C.c_int, [C.c_void_p, C.c_float]),
    "aae_ae_forward": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.POINTER(AaeRngInject), C.c_void_p]),
    "aae_output_layer_step": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p]),
    "aae_ae_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "aae_set_doc_l1": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aae_set_first_layer_external": (C.c_int, [C.c_void_p, C.c_int]),
    "aae_first_layer_forward": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch), C.c_void_p, C.c_void_p]),
    "aae_first_layer_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int, C.c_void_p]),
    "aae_apply_gathered": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]),
    "aae_cat_encode": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_int64, C.c_void_p]),
    "aae_cat_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int64, C.c_void_p]),
    "aae_csr_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                C.c_void_p, C.c_int64, C.c_void_p]),
    "aae_dense_to_csr": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.POINTER(C.c_int32 * 4), C.c_void_p]),
    "aae_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "aae_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "aae_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aae_join_output_layer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aae_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "aae_rccl_init": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(AaeCollectives)]),
    "aae_rccl_destroy": (C.c_int, [C.POINTER(AaeCollectives)]),
    "aae_dp_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(AaeCollectives), C.POINTER(AaeBatch), C.POINTER(AaeBatch),
                              C.POINTER(AaeBatch), C.c_void_p, C.POINTER(AaeRngInject), C.c_void_p]),
    "aae_dp_reserve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "aae_shard_step": (C.c_int, [C.c_void_p, C.POINTER(AaeCollectives), C.POINTER(AaeBatch), C.POINTER(AaeBatch), C.c_void_p,
                                 C.POINTER(AaeRngInject), C.c_float, C.c_void_p]),
    "aae_memcpy_sync": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "aae_echo_collectives": (C.c_int, [C.c_int32, C.POINTER(AaeCollectives)]),
    "aae_ipc_create": (C.c_int, [C.c_int64, C.c_char_p, C.POINTER(C.c_void_p)]),
    "aae_ipc_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(AaeCollectives)]),
    "aae_ipc_destroy": (C.c_int, [C.POINTER(AaeCollectives), C.c_void_p]),
    "aae_set_input_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "aae_prefetch_batch": (C.c_int, [C.c_void_p, C.POINTER(AaeBatch)]),
    "aae_set_split": (C.c_int, [C.c_void_p, C.c_int32]),
}
K_ENC_GATHER, K_DEC_BCE_FWD, K_DEC_DA2, K_DEC_DV3_ADAM, K_ENC_W1_ADAM, K_DEC_FUSED, K_CHAIN, K_DEC_CRIT, K_DEC_OPT, K_RANK, K_COLLECTIVE = range(11)

_lib = None


class AaeHipError(RuntimeError):
    pass


def _destroy_handle(lib, handle, arena, pid):
    # (a fork()ed child - e.g. a multiprocessing.Manager server started by the process that owns the model - inherits
    # the Python object but not a usable HIP runtime: only the creating process talks to the library)
    if os.getpid() == pid:
        try:
            lib.aae_destroy(handle)
        except Exception:
            pass
    del arena


def load_library():
    """dlopen the library and bind every symbol of the header.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the AAE step.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.aae_abi_version() != ABI_VERSION:
        raise ImportError("libaaerec_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise AaeHipError(f"libaaerec_hip error {rc}: {load_library().aae_last_error().decode()}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_UPLOAD_STREAMS = {}


def upload(x, device, dtype=None):
    """Host -> HBM copy of a small per-batch operand (condition inputs, index blocks) that does not drain the queue.

    A blocking copy from pageable memory on the compute stream waits for every kernel already enqueued there, so a
    training loop that uploads something per batch runs host and GPU in lock-step.  The copy goes through a side
    stream instead: the host waits for the copy alone, and because it has completed before this function returns,
    whatever is enqueued afterwards on the compute stream sees the data.  Device tensors pass through."""
    t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
    dev = torch.device(device)
    if t.is_cuda or dev.type != "cuda":
        return t.to(dev, dtype) if dtype is not None else t.to(dev)
    if dtype is not None:
        t = t.to(dtype)
    side = _UPLOAD_STREAMS.get(dev)
    if side is None:
        side = _UPLOAD_STREAMS[dev] = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        out = t.to(dev)
    out.record_stream(torch.cuda.current_stream(dev))
    return out


class HostRows:
    """The [n_rows, n_cols] float32 host matrix a predict loop hands back (the reference's predict() returns the
    dense score matrix, aae.py:840-870), filled batch by batch.  The matrix is allocated once in page-locked memory
    and each device batch is copied straight into its rows asynchronously - 56 GB/s on the MI355X host link against
    5 GB/s for batch.cpu().numpy() + np.vstack (three passes over pageable memory).  Above PIN_LIMIT bytes, or if
    pinning fails, the same scheme runs on a pageable matrix."""
    PIN_LIMIT = 32 << 30

    def __init__(self, n_rows, n_cols, device):
        self.device = torch.device(device)
        self.pinned = n_rows * n_cols * 4 <= self.PIN_LIMIT
        if self.pinned:
            try:
                self.host = torch.empty(n_rows, n_cols, dtype=torch.float32, pin_memory=True)
            except RuntimeError:
                self.pinned = False
        if not self.pinned:
            self.host = torch.empty(n_rows, n_cols, dtype=torch.float32)

    def put(self, start, batch):
        self.host[start:start + batch.shape[0]].copy_(batch, non_blocking=self.pinned)

    def numpy(self):
        if self.pinned:
            torch.cuda.current_stream(self.device).synchronize()
        return self.host.numpy()


def _cat_args(table, idx, block):
    """Argument checks shared by cat_encode / cat_update (operand shapes must match what the kernels index)."""
    if not (table.is_cuda and idx.is_cuda and block.is_cuda):
        raise RuntimeError("aaerec: the categorical condition kernels take GPU tensors (there is no CPU path)")
    if table.dtype != torch.float32 or not table.is_contiguous() or table.dim() != 2:
        raise TypeError("aaerec: embedding table must be a contiguous float32 [vocab, dim] tensor")
    if idx.dtype != torch.int32 or idx.dim() != 2 or not idx.is_contiguous():
        raise TypeError("aaerec: idx must be a contiguous int32 [rows, width] tensor")
    if block.dtype != torch.float32 or block.dim() != 2 or block.stride(1) != 1 or block.shape[0] != idx.shape[0] \
            or block.shape[1] != table.shape[1]:
        raise TypeError("aaerec: block must be a float32 [rows, dim] view with unit column stride")


def cat_encode(table, idx, out, mean=False):
    """CategoricalCondition.encode on the device: out[r] = sum (or mean over the padded width) of table[idx[r, w]];
    index 0 reads as zero.  `out` may be a column slice of the step's condition block."""
    _cat_args(table, idx, out)
    with torch.cuda.device(table.device):
        _check(load_library().aae_cat_encode(_ptr(table), table.shape[0], table.shape[1], _ptr(idx), idx.shape[0],
                                             idx.shape[1], CAT_MEAN if mean else CAT_SUM, _ptr(out), out.stride(0),
                                             C.c_void_p(torch.cuda.current_stream(table.device).cuda_stream)))


def cat_update(table, exp_avg, exp_avg_sq, idx, dout, lr, step, mean=False, grad_scratch=None):
    """Backward of cat_encode from dout [rows, dim] plus the condition's optimiser step: SparseAdam over the rows the
    batch names, or (grad_scratch given: zeros [vocab, dim]) dense Adam over the whole table.  step counts from 1."""
    _cat_args(table, idx, dout)
    for t in (exp_avg, exp_avg_sq) + ((grad_scratch,) if grad_scratch is not None else ()):
        if t.shape != table.shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != table.device:
            raise TypeError("aaerec: optimiser state must match the embedding table")
    with torch.cuda.device(table.device):
        _check(load_library().aae_cat_update(_ptr(table), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(grad_scratch),
                                             table.shape[0], table.shape[1], _ptr(idx), idx.shape[0], idx.shape[1],
                                             CAT_MEAN if mean else CAT_SUM, _ptr(dout), dout.stride(0),
                                             CAT_ADAM if grad_scratch is not None else CAT_SPARSE_ADAM, float(lr), int(step),
                                             C.c_void_p(torch.cuda.current_stream(table.device).cuda_stream)))


def csr_embed(csr, table):
    """[rows, dim] device tensor: row r = sum of csr.values[e] * table[csr.indices[e]] over the entries of row r
    (EmbeddedVectorizer.transform's `sparse_scores @ embedding`, reference ub.py:52-57)."""
    if not table.is_cuda or table.dtype != torch.float32 or table.dim() != 2 or table.stride(1) != 1:
        raise TypeError("aaerec: table must be a float32 [vocab, dim] GPU tensor with unit column stride")
    if csr.shape[1] > table.shape[0]:
        raise ValueError("aaerec: the sparse matrix has more columns than the table has rows")
    out = torch.empty(csr.shape[0], table.shape[1], dtype=torch.float32, device=table.device)
    with torch.cuda.device(table.device):
        _check(load_library().aae_csr_embed(_ptr(csr.indptr), _ptr(csr.indices), _ptr(csr.values), csr.shape[0],
                                            _ptr(table), table.shape[0], table.shape[1], table.stride(0), _ptr(out),
                                            out.stride(0),
                                            C.c_void_p(torch.cuda.current_stream(table.device).cuda_stream)))
    return out


_GENERATION = itertools.count(1)


def row_ids(rows, generation):
    """Tag a device int32 row-id tensor with the caller's content id (aae_batch.generation, ABI 3): the same id for every
    view of one buffer while its content stands, a new one after any rewrite or re-allocation.  A batch whose row ids carry
    no tag is never matched with work built ahead for it (aae_prefetch_batch): a slice of a tensor is a new Python object at
    an address the allocator may have handed out before, and nothing but the caller knows whether the ids in it changed."""
    rows._aae_generation = int(generation)
    return rows


class DeviceCSR:
    """A CSR matrix resident in HBM (int64 indptr, int32 indices, float32 values).  `generation` names its content
    (aae_batch.generation): unique per object; call touch() after rewriting the arrays in place."""

    def touch(self):
        self.generation = next(_GENERATION)

    def __init__(self, X, device):
        X = X.tocsr()
        X.sum_duplicates()
        self.generation = next(_GENERATION)
        self.shape = X.shape
        self.nnz_per_row_max = int(np.diff(X.indptr).max()) if X.shape[0] else 0
        self.indptr = upload(X.indptr.astype(np.int64), device)
        # (a matrix without entries - e.g. an item slice none of the batch's documents touches - still hands the
        # library valid pointers: one unused element)
        self.indices = upload(X.indices.astype(np.int32) if X.nnz else np.zeros(1, dtype=np.int32), device)
        self.values = upload(X.data.astype(np.float32) if X.nnz else np.zeros(1, dtype=np.float32), device)

    @classmethod
    def from_dense(cls, X, device, capacity):
        """The reference's call form (aae.py:745-754): a dense [rows, n_cols] float32 / float64 batch.  The matrix is
        uploaded as it is (one pass of host memory: no host-side scan, no float64 -> float32 copy) and compacted on the
        device (aae_dense_to_csr); raises like the reference's BCE for values outside [0, 1]."""
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float32)
        if X.ndim != 2:
            raise ValueError("expected a 2-D batch")
        dev = torch.device(device)
        rows, n_cols = X.shape
        dense = upload(X, dev)
        self = cls.__new__(cls)
        self.generation = next(_GENERATION)
        self.shape = (rows, n_cols)
        cap = int(max(1, min(int(capacity), rows * n_cols)))
        self.indptr = torch.empty(rows + 1, dtype=torch.int64, device=dev)
        self.indices = torch.empty(cap, dtype=torch.int32, device=dev)
        self.values = torch.empty(cap, dtype=torch.float32, device=dev)
        scratch = torch.empty(rows + 8, dtype=torch.int32, device=dev)
        stats = (C.c_int32 * 4)()
        with torch.cuda.device(dev):
            _check(load_library().aae_dense_to_csr(_ptr(dense), X.dtype.itemsize, n_cols, rows, n_cols, _ptr(self.indptr),
                                                   _ptr(self.indices), _ptr(self.values), cap, _ptr(scratch), C.byref(stats),
                                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if stats[2]:
            raise RuntimeError("all elements of target should be between 0 and 1")
        if stats[3]:
            raise ValueError(f"the batch holds {stats[1]} entries, more than the model's capacity of {cap} (max_nnz)")
        self.nnz_per_row_max = int(stats[0])
        self.nnz = int(stats[1])
        return self

    @classmethod
    def from_arrays(cls, indptr, indices, values, n_cols, device):
        self = cls.__new__(cls)
        self.generation = next(_GENERATION)
        self.shape = (len(indptr) - 1, n_cols)
        self.nnz_per_row_max = int(np.diff(indptr).max()) if len(indptr) > 1 else 0
        self.indptr = upload(np.asarray(indptr, dtype=np.int64), device)
        self.indices = upload(np.asarray(indices, dtype=np.int32) if len(indices) else np.zeros(1, dtype=np.int32), device)
        self.values = upload(np.asarray(values, dtype=np.float32) if len(values) else np.zeros(1, dtype=np.float32), device)
        return self


# state_dict key <-> (net, layer)
_NETS = {"enc": 0, "dec": 1, "disc": 2}
_PARAM_ID = {("enc", 1): T_ENC_W1T, ("enc", 2): T_ENC_W2, ("enc", 3): T_ENC_W3,
             ("dec", 1): T_DEC_V1, ("dec", 2): T_DEC_V2, ("dec", 3): T_DEC_V3,
             ("disc", 1): T_DISC_D1, ("disc", 2): T_DISC_D2, ("disc", 3): T_DISC_D3}


class HipAAE:
    """One model replica on one GPU: owns the arena and the C handle."""

    def __init__(self, n_items, n_hidden, n_code, cond_inc=0, max_batch=100, max_nnz=None,
                 activation="ReLU", prior="gauss", prior_scale=None, optimizer="adam",
                 normalize_inputs=True, dropout=(.2, .2), gen_lr=1e-3, reg_lr=1e-3,
                 rng_mode="device", seed=0, grad_mode="fused", device=None, unfused_decoder=False,
                 dp_world=1, w1_cap=None, ae_only=False, vae=False, dtype="f32", blocked_output=False, dense_noise=False):
        lib = load_library()
        if not torch.cuda.is_available():
            raise AaeHipError("no HIP device: the AAE step has no CPU fallback")
        if activation not in ACTIVATIONS:
            raise ValueError(f"activation {activation!r} has no gfx950 kernel (supported: {sorted(ACTIVATIONS)})")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.lib = lib
        cfg = AaeConfig()
        cfg.abi_version = ABI_VERSION
        cfg.n_items, cfg.n_hidden, cfg.n_code, cfg.cond_inc = n_items, n_hidden, n_code, cond_inc
        cfg.max_batch = max_batch
        # capacity of the per-batch scratch lists; the default allows 4096 entries per row
        cfg.max_nnz = int(max_nnz if max_nnz is not None else min(2 ** 31 - 1, max_batch * min(n_items, 4096)))
        cfg.activation = ACTIVATIONS[activation]
        cfg.enc_final = FINALS[{"gauss": "linear", "categorical": "softmax", "bernoulli": "sigmoid"}[prior]]
        cfg.optimizer = OPTIMIZERS[optimizer]
        cfg.normalize_inputs = int(bool(normalize_inputs))
        cfg.rng_mode = RNG_DEVICE if rng_mode == "device" else RNG_INJECT
        cfg.prior = PRIORS[prior]
        cfg.grad_mode = GRAD_EXPORT if grad_mode == "export" else GRAD_FUSED
        cfg.dropout1, cfg.dropout2 = float(dropout[0]), float(dropout[1])
        cfg.gen_lr, cfg.reg_lr = float(gen_lr), float(reg_lr)
        cfg.has_prior_scale = int(prior_scale is not None)
        cfg.prior_scale = float(prior_scale) if prior_scale is not None else 1.0
        cfg.seed = int(seed) & (2 ** 64 - 1)
        cfg.unfused_decoder = 1 if unfused_decoder else 0
        cfg.dp_world = int(dp_world) if grad_mode == "export" else 0
        cfg.model_kind = MODEL_VAE if vae else MODEL_AE if ae_only else MODEL_AAE
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        cfg.dtype = DTYPE_BF16 if dtype == "bf16" else DTYPE_F32
        # batches beyond one fused launch's 112 rows as row-blocked launches of the fused output layer (up to 1664 rows;
        # DESIGN.md 3.1) instead of the three streaming GEMMs: what fit() asks for at such batch sizes and the item
        # slices of the sharded schemes always.  Both dtypes since late r4 (bf16 on the rounded-operand form of the same
        # kernels: C3's shape at batch 512 1.01 -> 0.56 ms/step, C4 0.50 -> 0.40)
        cfg.blocked_output = 1 if (blocked_output and grad_mode != "export") else 0
        # DenoisingAutoEncoder(corrupt='gauss'): room for the dense noisy encoder input (set_input_noise before a step)
        cfg.dense_noise = 1 if dense_noise else 0
        self.dtype = dtype
        self.ae_only, self.vae = bool(ae_only or vae), bool(vae)
        self.dp_world = int(dp_world)
        # rows of the packed first-layer gradient one rank may send per exchange
        self.w1_cap = int(w1_cap if w1_cap is not None else min(cfg.max_nnz, n_items, 65536))
        self._w1_hdr = (1 + self.w1_cap + 3) & ~3          # int32 words of the packet header (16-byte aligned)
        self._w1_packet = None
        self.cfg = cfg
        self.N, self.h, self.c, self.cond_inc = n_items, n_hidden, n_code, cond_inc
        self.max_batch = max_batch
        nbytes = C.c_size_t()
        _check(lib.aae_arena_bytes(C.byref(cfg), C.byref(nbytes)))
        with self._on_device():
            self.arena = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
            assert self.arena.data_ptr() % 256 == 0
            h = C.c_void_p()
            _check(lib.aae_create(C.byref(cfg), C.c_void_p(self.arena.data_ptr()), nbytes.value, self._stream(),
                                  C.byref(h)))
        self.handle = h
        # aae_destroy waits for the handle's side stream (the deferred dec_optim launch of the last step reads and writes
        # the arena): it must run BEFORE the arena's memory can go back to the driver - also at interpreter exit, where
        # __del__ is not guaranteed to.  The finalizer owns a reference to the arena, so the order is fixed.
        self._finalizer = weakref.finalize(self, _destroy_handle, lib, h, self.arena, os.getpid())
        _check(lib.aae_set_lr(self.handle, float(gen_lr), float(reg_lr)))
        self._keep = []   # device buffers of the running step
        self._ga1_ld = None
        self._ext_first = False
        self._rank_rows = {}

    def close(self):
        """Destroy the handle now (waits for its side stream)."""
        self._finalizer()
        self.handle = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        """Context with this handle's device current.  One process per GPU is the rule: the device is almost always
        current already, and torch.cuda.device()'s enter / exit (two runtime calls and a Python context per library call,
        ~30 calls per data-parallel step) is skipped then."""
        if torch.cuda.current_device() == self._dev_index:
            return _NULL_CTX
        return torch.cuda.device(self.device)

    # ---- views into the arena ---------------------------------------------------------
    def tensor(self, tid, padded=False):
        """float32 view [rows, cols] (strided by ld) of a tensor of the model.  (The current stream first waits for the
        deferred optimiser launch of the last step, if one is pending: a view is about to read or write what it touches.)"""
        if tid >= T_ACT_Z:          # activations: only the deferred optimiser launch reads them (a prefetch does not)
            _check(self.lib.aae_join_output_layer(self.handle, self._stream()))
        else:
            self.join()
        info = AaeTensor()
        _check(self.lib.aae_tensor_info(self.handle, tid, C.byref(info)))
        flat = self.arena[info.byte_offset: info.byte_offset + info.rows * info.ld * 4].view(torch.float32)
        full = flat.view(info.rows, info.ld)
        return full if padded else full[:, :info.cols]

    def _span(self, tid_lo, tid_hi):
        """One flat float32 view from the start of tensor tid_lo to the end of tid_hi (they are
        adjacent in the arena; the 256-byte alignment gaps in between are never written)."""
        self.join()
        a, b = AaeTensor(), AaeTensor()
        _check(self.lib.aae_tensor_info(self.handle, tid_lo, C.byref(a)))
        _check(self.lib.aae_tensor_info(self.handle, tid_hi, C.byref(b)))
        end = b.byte_offset + b.rows * b.ld * 4
        return self.arena[a.byte_offset:end].view(torch.float32)

    def grad_buckets(self, which):
        """Flat gradient views to all-reduce for optimiser `which` (export mode).  The first
        encoder layer's row-sparse gradient is not among them: see w1_export / w1_import."""
        if which in (O_ENC, O_GEN):
            return []        # b1, W2, W3 ride in the w1_export packet and are applied by w1_import
        if which == O_DEC:
            return [self._span(T_GRAD + T_DEC_V1, T_GRAD + T_DEC_V2), self._span(T_GRAD + T_DEC_V3, T_GRAD + T_DEC_V3)]
        if which == "dec_small":
            return [self._span(T_GRAD + T_DEC_V1, T_GRAD + T_DEC_V2)]
        if which == "enc_small":        # (first layer external: its bias, W2, W3)
            return [self._span(T_GRAD + T_ENC_B1, T_GRAD + T_ENC_W3)]
        if which == "enc_dec_small":    # b1, W2, W3, V1, V2: one contiguous span of the arena
            return [self._span(T_GRAD + T_ENC_B1, T_GRAD + T_DEC_V2)]
        return [self._span(T_GRAD + T_DISC_D1, T_GRAD + T_DISC_D3)]

    def params_changed(self):
        """Parameters were written through arena views: derived copies are rebuilt before their next use."""
        _check(self.lib.aae_params_changed(self.handle))

    def join(self):
        """Make the current stream wait for the last step's deferred dec_optim launch (aae_join; no-op when none)."""
        _check(self.lib.aae_join(self.handle, self._stream()))

    def set_split(self, workgroups):
        """0: the fused output layer as one launch on the caller's stream; n > 0: split form, n workgroups for the
        deferred dec_optim launch (aae_set_split)."""
        self.sync()
        _check(self.lib.aae_set_split(self.handle, int(workgroups)))

    # ---- state_dict in the reference layout ------------------------------------------
    def sync(self):
        """Replay the deferred W1T updates so that arena views show eager-equivalent values."""
        with self._on_device():
            _check(self.lib.aae_sync(self.handle, self._stream()))

    def load_params(self, params):
        """params: {'enc.lin1.weight': ndarray [out,in], 'enc.lin1.bias': ...} (any subset of layers,
        weight and bias together)."""
        self.sync()
        for (net, layer), tid in _PARAM_ID.items():
            wk, bk = f"{net}.lin{layer}.weight", f"{net}.lin{layer}.bias"
            if wk not in params:
                continue
            w = torch.as_tensor(np.asarray(params[wk], dtype=np.float32), device=self.device)
            b = torch.as_tensor(np.asarray(params[bk], dtype=np.float32), device=self.device)
            self._put(tid, w, b)
        torch.cuda.synchronize(self.device)
        _check(self.lib.aae_params_changed(self.handle))     # (written through arena views, not aae_load_linear)

    def _put(self, tid, w, b, tid_b1=T_ENC_B1):
        if self._is_w1t(tid):
            self.tensor(tid).copy_(w.t())
            self.tensor(tid_b1)[0].copy_(b)
        else:
            t = self.tensor(tid)
            t[:, :-1].copy_(w)
            t[:, -1].copy_(b)

    @staticmethod
    def _is_w1t(tid):
        return tid in (T_ENC_W1T, T_ADAM_ENC, T_ADAM_ENC + 1, T_ADAM_GEN, T_ADAM_GEN + 1)

    def _get(self, tid, tid_b1):
        if self._is_w1t(tid):
            return self.tensor(tid).t().contiguous().cpu().numpy(), self.tensor(tid_b1)[0].cpu().numpy().copy()
        t = self.tensor(tid)
        return t[:, :-1].contiguous().cpu().numpy(), t[:, -1].contiguous().cpu().numpy()

    def state_dict(self):
        self.sync()
        out = {}
        for (net, layer), tid in _PARAM_ID.items():
            w, b = self._get(tid, T_ENC_B1)
            out[f"{net}.lin{layer}.weight"], out[f"{net}.lin{layer}.bias"] = w, b
        return out

    def adam_state(self, which):
        """{'lin1.weight': (m, v), ...} of optimiser `which` in 'enc','dec','gen','disc', + step."""
        base, net, lo = {"enc": (T_ADAM_ENC, "enc", 0), "gen": (T_ADAM_GEN, "enc", 0),
                         "dec": (T_ADAM_DEC, "dec", 0), "disc": (T_ADAM_DISC, "disc", 0)}[which]
        self.sync()
        out = {}
        if net == "enc":
            slots = {1: 0, 2: 2, 3: 3}
            b1m, b1v = base + 2, base + 3
        else:
            slots = {1: 0, 2: 1, 3: 2}
        for layer, slot in slots.items():
            tm, tv = base + 2 * slot, base + 2 * slot + 1
            if net == "enc" and layer == 1:
                mw, mb = self._get(tm, b1m)
                vw, vb = self._get(tv, b1v)
            else:
                mw, mb = self._get(tm, None)
                vw, vb = self._get(tv, None)
            out[f"lin{layer}.weight"] = (mw, vw)
            out[f"lin{layer}.bias"] = (mb, vb)
        step = C.c_int64()
        oid = {"enc": O_ENC, "dec": O_DEC, "gen": O_GEN, "disc": O_DISC}[which]
        _check(self.lib.aae_store_adam(self.handle, oid, 2, None, None, None, None, C.byref(step)))
        out["step"] = step.value
        return out

    # ---- batches / randomness ----------------------------------------------------------
    def _batch(self, csr, row_start, n_rows, rows=None, bounded=True):
        b = AaeBatch()
        b.indptr_dev, b.indices_dev, b.values_dev = csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr()
        b.rows_dev = rows.data_ptr() if rows is not None else None
        b.row_start, b.n_rows = int(row_start), int(n_rows)
        b.nnz_bound = int(n_rows * max(1, csr.nnz_per_row_max))
        if bounded and b.nnz_bound > self.cfg.max_nnz:
            raise ValueError(f"batch may hold {b.nnz_bound} entries but the model was created with max_nnz="
                             f"{self.cfg.max_nnz}")
        b.max_row_nnz = int(csr.nnz_per_row_max)
        # the content id a named-ahead batch is matched by (ABI 3): the matrix's generation and the row ids' (row_ids());
        # 0 - never matched - when either is unknown
        cg = int(getattr(csr, "generation", 0))
        rg = 1 if rows is None else int(getattr(rows, "_aae_generation", 0))
        b.generation = ((cg << 32) ^ rg) & 0x7FFFFFFFFFFFFFFF if cg and rg else 0
        return b

    def _inject(self, masks, z_real):
        # every step-opening entry point passes through here: the buffers kept alive for the PREVIOUS step (condition
        # blocks, injected masks, packets handed to the library by pointer) are released now, whatever the rng mode
        self._keep = []
        if masks is None and z_real is None:
            return None
        inj = AaeRngInject()
        keep = []
        for i in range(12):
            mk = None if masks is None or i >= len(masks) or masks[i] is None else masks[i]
            if mk is not None:
                t = torch.as_tensor(np.ascontiguousarray(mk, dtype=np.uint8), device=self.device) \
                    if not torch.is_tensor(mk) else mk.to(self.device, torch.uint8).contiguous()
                keep.append(t)
                inj.masks_dev[i] = t.data_ptr()
            else:
                inj.masks_dev[i] = None
        if z_real is not None:
            z = torch.as_tensor(np.ascontiguousarray(z_real, dtype=np.float32), device=self.device) \
                if not torch.is_tensor(z_real) else z_real.to(self.device, torch.float32).contiguous()
            keep.append(z)
            inj.z_real_dev = z.data_ptr()
        self._keep = keep
        return inj

    # ---- the step --------------------------------------------------------------------
    def prefetch(self, csr, row_start, n_rows, rows=None):
        """Name the batch of the step AFTER the next one (aae_prefetch_batch): its unique-item list and deferred-Adam
        catch-up run beside the next step.  `rows` (device int32 row ids) must stay alive and unchanged until then."""
        b = self._batch(csr, row_start, n_rows, rows)
        self._pf_keep = (csr, rows)
        _check(self.lib.aae_prefetch_batch(self.handle, C.byref(b)))

    def set_input_noise(self, noise):
        """The NEXT training step's encoder reads the dense batch + `noise` ([rows, n_items] float32, already scaled by
        the noise factor) instead of the sparse rows (aae_set_input_noise; DenoisingAutoEncoder corrupt='gauss')."""
        noise = noise.to(self.device, torch.float32).contiguous()
        assert noise.shape[1] == self.N
        self._noise_keep = noise
        _check(self.lib.aae_set_input_noise(self.handle, _ptr(noise), noise.shape[1]))

    def step(self, csr, row_start, n_rows, rows=None, cond=None, masks=None, z_real=None):
        """One partial_fit without generic conditions (cond: device tensor [n_rows, cond_inc])."""
        b = self._batch(csr, row_start, n_rows, rows)
        inj = self._inject(masks, z_real)
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous()
            self._keep.append(cond)
        with self._on_device():
            _check(self.lib.aae_step(self.handle, C.byref(b), _ptr(cond), C.byref(inj) if inj else None,
                                     self._stream()))

    # ---- the ae phase cut at the decoder's output layer (vocabulary-sharded data parallelism) -----------
    def ae_forward(self, csr, row_start, n_rows, rows=None, cond=None, masks=None, z_real=None):
        """Encoder + decoder hidden layers; the last hidden activation stays in tensor(T_ACT_DH2, padded=True)."""
        b = self._batch(csr, row_start, n_rows, rows)
        inj = self._inject(masks, z_real)
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous()
            self._keep.append(cond)
        with self._on_device():
            _check(self.lib.aae_ae_forward(self.handle, C.byref(b), _ptr(cond), C.byref(inj) if inj else None,
                                           self._stream()))

    def output_layer_step(self, csr=None, row_start=0, n_rows=0, rows=None):
        """The decoder's output layer over this handle's items from T_ACT_DH2 -> dL/d(dh2) in T_ACT_DA2; csr = None
        continues the step ae_forward started on this handle."""
        b = self._batch(csr, row_start, n_rows, rows) if csr is not None else None
        with self._on_device():
            _check(self.lib.aae_output_layer_step(self.handle, C.byref(b) if b is not None else None, self._stream()))

    def ae_backward(self, da2=None):
        """Decoder hidden + encoder backward from dL/d(dh2) (None = this handle's T_ACT_DA2)."""
        if da2 is not None:
            assert da2.is_cuda and da2.dtype == torch.float32 and da2.stride(1) == 1
            self._keep.append(da2)
        with self._on_device():
            _check(self.lib.aae_ae_backward(self.handle, _ptr(da2), da2.stride(0) if da2 is not None else 0,
                                            self._stream()))

    # ---- the first encoder layer sharded over the vocabulary as well (include/aaerec_hip.h: aae_first_layer_*) ------
    def set_doc_l1(self, doc_l1):
        """Slice handle: L1 norms of the complete documents (float32 device tensor indexed by document id; None = the
        batch's own rows are complete)."""
        if doc_l1 is not None:
            assert doc_l1.is_cuda and doc_l1.dtype == torch.float32 and doc_l1.is_contiguous()
        self._doc_l1 = doc_l1                      # (keeps the buffer alive as long as the handle uses it)
        _check(self.lib.aae_set_doc_l1(self.handle, _ptr(doc_l1)))

    def set_first_layer_external(self, on=True):
        """Replica handle: a1_rows() is filled by the caller, dL/d(a1) comes back in ga1_rows(); enc.lin1 stays untouched."""
        _check(self.lib.aae_set_first_layer_external(self.handle, 1 if on else 0))
        self._ext_first = bool(on)

    def ga1_packet(self, n_rows, with_decoder):
        """(flat view, rows part in floats, offset of the small layers' gradient span in floats): dL/d(a1) of the last
        encoder backward and, right behind it in the arena, the gradients of enc.lin1's bias, enc.lin2, enc.lin3 (and
        dec.lin1, dec.lin2 when with_decoder) - ONE contiguous packet for the both-sharded scheme's all-gather.  Export
        mode with an external first layer only (the library keeps dL/d(a1) right-aligned in front of that span)."""
        a, b, e = AaeTensor(), AaeTensor(), AaeTensor()
        _check(self.lib.aae_tensor_info(self.handle, T_ACT_GA1, C.byref(a)))
        _check(self.lib.aae_tensor_info(self.handle, T_GRAD + T_ENC_B1, C.byref(b)))
        _check(self.lib.aae_tensor_info(self.handle, T_GRAD + (T_DEC_V2 if with_decoder else T_ENC_W3), C.byref(e)))
        start = a.byte_offset + (a.rows - n_rows) * a.ld * 4
        end = e.byte_offset + e.rows * e.ld * 4
        assert self._ext_first and start + n_rows * a.ld * 4 <= b.byte_offset < end
        return self.arena[start:end].view(torch.float32), n_rows * a.ld, (b.byte_offset - start) // 4

    def first_layer_forward(self, csr=None, row_start=0, n_rows=0, rows=None, bias=None):
        """This handle's items' share of the first layer's pre-activations -> a1_rows(); csr = None: the running batch
        again with the weights as they are now.  bias: the replicas' enc.lin1 bias (device tensor) on exactly one share."""
        b = self._batch(csr, row_start, n_rows, rows) if csr is not None else None
        with self._on_device():
            _check(self.lib.aae_first_layer_forward(self.handle, C.byref(b) if b is not None else None, _ptr(bias),
                                                    self._stream()))

    def first_layer_update(self, which, ga1=None, rows_per_block=0, block_stride=0):
        """enc.lin1's rows of this handle's items from dL/d(a1) of the running batch, optimiser `which`.  ga1 = None:
        ga1_rows(); else a float32 device tensor holding blocks of `rows_per_block` rows (leading dimension = ga1_rows()'s),
        `block_stride` floats apart (the ranks' packets of an all-gather, read where they landed)."""
        if ga1 is not None:         # (the caller's buffer: it must stay alive until the stream has passed this call)
            assert ga1.is_cuda and ga1.dtype == torch.float32 and ga1.is_contiguous()
        if ga1 is not None and self._ga1_ld is None:
            self._ga1_ld = self.ga1_rows(1).stride(0)
        ld = self._ga1_ld if ga1 is not None else 0
        with self._on_device():
            _check(self.lib.aae_first_layer_update(self.handle, _ptr(ga1), ld, int(rows_per_block), int(block_stride),
                                                   int(which), self._stream()))

    def first_layer_bias(self):
        """enc.lin1's bias as a device view (what one share of first_layer_forward adds)."""
        return self.tensor(T_ENC_B1, padded=True)

    def apply_gathered(self, which_a, which_b, packets, peer_stride, n_peers, span_offset):
        """Optimiser which_a (+ dec_optim's small layers when which_b == O_DEC, else -1) on the replica's small layers
        from the gathered packets (aae_apply_gathered): one launch, the peers summed in rank order inside it."""
        with self._on_device():
            _check(self.lib.aae_apply_gathered(self.handle, int(which_a), int(which_b), _ptr(packets), int(peer_stride),
                                               int(n_peers), int(span_offset), self._stream()))

    def dp_reserve(self, n_rows, world):
        """Set-up for dp_step: the gathered-packet scratch for `world` ranks x n_rows local documents (aae_dp_reserve)."""
        with self._on_device():
            _check(self.lib.aae_dp_reserve(self.handle, int(n_rows), int(world)))

    def dp_step(self, slice_model, coll, csr, row_start, n_rows, slice_csr, g_row_start, global_rows, rows=None, g_rows=None,
                next_global=None, cond=None, masks=None, z_real=None):
        """One data-parallel partial_fit as ONE library call (aae_dp_step): this replica's documents + the global batch in
        the slice model's corpus; every kernel and every collective of the step is enqueued by the library on this
        handle's stream.  coll: an AaeCollectives table (parallel.rccl_collectives / parallel.python_collectives).
        next_global = (row_start, rows) of the NEXT global batch in slice_csr: named ahead to the slice model."""
        b = self._batch(csr, row_start, n_rows, rows)
        g = slice_model._batch(slice_csr, g_row_start, global_rows, g_rows)
        nxt = None
        if next_global is not None:
            nxt = slice_model._batch(slice_csr, next_global[0], global_rows, next_global[1])
            slice_model._pf_keep = (slice_csr, next_global[1])
        inj = self._inject(masks, z_real)
        slice_model._keep = []
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous()
            self._keep.append(cond)
        self._keep.append((g_rows, rows))
        with self._on_device():
            _check(self.lib.aae_dp_step(self.handle, slice_model.handle, C.byref(coll), C.byref(b), C.byref(g),
                                        C.byref(nxt) if nxt is not None else None, _ptr(cond),
                                        C.byref(inj) if inj else None, self._stream()))

    def shard_step(self, coll, csr, row_start, n_rows, item_share, rows=None, next_rows=None, cond=None, masks=None, z_real=None):
        """One partial_fit of the item-sharded model with replicated hidden stacks as ONE library call (aae_shard_step):
        this handle = an item slice of enc.lin1 / dec.lin3 + a full copy of the hidden layers; the batch is the GLOBAL batch
        in the slice's corpus; three all-reduces of [rows, n_hidden] partial sums through `coll`.  next_rows = (row_start,
        rows, n_rows) of the next global batch (named ahead; rows = device int32 row ids or None)."""
        b = self._batch(csr, row_start, n_rows, rows)
        nxt = None
        if next_rows is not None:
            nxt = self._batch(csr, next_rows[0], next_rows[2], next_rows[1])
            self._pf_keep = (csr, next_rows[1])
        inj = self._inject(masks, z_real)
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous()
            self._keep.append(cond)
        self._keep.append(rows)
        with self._on_device():
            _check(self.lib.aae_shard_step(self.handle, C.byref(coll), C.byref(b), C.byref(nxt) if nxt is not None else None,
                                           _ptr(cond), C.byref(inj) if inj else None, float(item_share), self._stream()))

    def a1_rows(self, n_rows):
        """[n_rows, ld] view of the first layer's pre-activations."""
        return self.tensor(T_ACT_A1, padded=True)[:n_rows]

    def ga1_rows(self, n_rows):
        """[n_rows, ld] view of dL/d(a1) of the last encoder backward (a replica in export mode with an external first
        layer keeps it in the LAST rows of its buffer: see ga1_packet)."""
        t = self.tensor(T_ACT_GA1, padded=True)
        return t[t.shape[0] - n_rows:] if (self._ext_first and self.cfg.grad_mode == GRAD_EXPORT) else t[:n_rows]

    def dh2_rows(self, n_rows):
        """[n_rows, ld] view of the decoder's last hidden activation (with its bias-input column and padding)."""
        return self.tensor(T_ACT_DH2, padded=True)[:n_rows]

    def da2_rows(self, n_rows):
        """[n_rows, ld] view of dL/d(dh2): written by output_layer_step, read by ae_backward()."""
        return self.tensor(T_ACT_DA2, padded=True)[:n_rows]

    def cond_grad(self, n_rows):
        """dL/d(cond) of the last step's autoencoder phase: [n_rows, cond_inc] view (trainable conditions)."""
        return self.tensor(T_ACT_DZC)[:n_rows, self.c:]

    def ae_encode(self, csr, row_start, n_rows, rows=None, masks=None, z_real=None):
        b = self._batch(csr, row_start, n_rows, rows)
        inj = self._inject(masks, z_real)
        z = torch.empty(n_rows, self.c, dtype=torch.float32, device=self.device)
        with self._on_device():
            _check(self.lib.aae_ae_encode(self.handle, C.byref(b), C.byref(inj) if inj else None, _ptr(z),
                                          self._stream()))
        return z

    def ae_decode_backward(self, zc):
        zc = zc.detach().to(self.device, torch.float32).contiguous()
        assert zc.shape[1] == self.c + self.cond_inc, "conditions.size_increment() mismatch"
        dzc = torch.empty_like(zc)
        with self._on_device():
            _check(self.lib.aae_ae_decode_backward(self.handle, _ptr(zc), zc.shape[1], None, _ptr(dzc), self._stream()))
        return dzc

    def decoder_step(self, csr, row_start, n_rows, zin, rows=None, masks=None, want_grad=True):
        """DecodingRecommender.partial_fit (aae.py:489-517): decoder forward on zin [n_rows, n_code + cond_inc],
        BCE against the CSR rows, decoder backward + optimiser step.  masks: (dec.drop1, dec.drop2) keep-masks
        in rng_mode='inject'.  Returns dL/dzin (device) or None."""
        b = self._batch(csr, row_start, n_rows, rows)
        inj = self._inject([None, None, masks[0], masks[1]] if masks is not None else None, None)
        zin = zin.detach().to(self.device, torch.float32).contiguous()
        assert zin.shape == (n_rows, self.c + self.cond_inc), "decoder input width mismatch"
        dz = torch.empty_like(zin) if want_grad else None
        self._keep.append(zin)
        with self._on_device():
            _check(self.lib.aae_decoder_step(self.handle, C.byref(b), _ptr(zin), zin.shape[1],
                                             C.byref(inj) if inj else None, _ptr(dz), self._stream()))
        return dz

    def vae_step(self, csr, row_start, n_rows, rows=None, cond=None, eps=None):
        """VAE.partial_fit (vae.py:147-186).  eps: [n_rows, n_code] standard-normal draws (rng_mode='inject')."""
        b = self._batch(csr, row_start, n_rows, rows)
        keep = []
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous(); keep.append(cond)
        if eps is not None:
            eps = torch.as_tensor(eps, dtype=torch.float32).to(self.device).contiguous(); keep.append(eps)
        self._keep = keep
        with self._on_device():
            _check(self.lib.aae_vae_step(self.handle, C.byref(b), _ptr(cond), _ptr(eps), self._stream()))

    def vae_encode(self, csr, row_start, n_rows, rows=None, eps=None, train=True):
        """z [n_rows, n_code] of the VAE's encoder + reparametrisation (aae_vae_encode); train=True opens a step."""
        b = self._batch(csr, row_start, n_rows, rows)
        if eps is not None:
            eps = torch.as_tensor(eps, dtype=torch.float32).to(self.device).contiguous()
        z = torch.empty(n_rows, self.c, dtype=torch.float32, device=self.device)
        self._keep = [eps, z]
        with self._on_device():
            _check(self.lib.aae_vae_encode(self.handle, C.byref(b), _ptr(eps), _ptr(z), int(bool(train)), self._stream()))
        return z

    def vae_decode_backward(self, zc):
        zc = zc.detach().to(self.device, torch.float32).contiguous()
        dzc = torch.empty_like(zc)
        self._keep = self._keep + [zc, dzc]
        with self._on_device():
            _check(self.lib.aae_vae_decode_backward(self.handle, _ptr(zc), zc.shape[1], _ptr(dzc), self._stream()))
        return dzc

    def vae_encoder_backward(self, dz):
        dz = dz.detach().to(self.device, torch.float32).contiguous()
        self._keep = self._keep + [dz]
        with self._on_device():
            _check(self.lib.aae_vae_encoder_backward(self.handle, _ptr(dz), dz.shape[1], self._stream()))

    def vae_predict(self, csr, row_start, n_rows, cond=None, eps=None):
        b = self._batch(csr, row_start, n_rows)
        out = self._out_buffer(n_rows)
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous()
        if eps is not None:
            eps = torch.as_tensor(eps, dtype=torch.float32).to(self.device).contiguous()
        with self._on_device():
            _check(self.lib.aae_vae_predict(self.handle, C.byref(b), _ptr(cond), _ptr(eps), _ptr(out), out.shape[1],
                                            self._stream()))
        return out[:, :self.N]

    def ae_encoder_backward(self, dz):
        dz = dz.detach().to(self.device, torch.float32).contiguous()
        self._keep.append(dz)
        with self._on_device():
            _check(self.lib.aae_ae_encoder_backward(self.handle, _ptr(dz), dz.shape[1], self._stream()))

    def disc_gen(self):
        with self._on_device():
            _check(self.lib.aae_disc_gen(self.handle, None, self._stream()))

    def disc_step(self):
        with self._on_device():
            _check(self.lib.aae_disc_step(self.handle, None, self._stream()))

    def gen_step(self):
        with self._on_device():
            _check(self.lib.aae_gen_step(self.handle, None, self._stream()))

    def apply_updates(self, which, skip=-1):
        with self._on_device():
            _check(self.lib.aae_apply_updates_except(self.handle, which, skip, self._stream()))

    def apply_shard(self, tid, row_begin, row_end, grad_shard, which):
        self._keep.append(grad_shard)
        with self._on_device():
            _check(self.lib.aae_apply_shard(self.handle, tid, row_begin, row_end, C.c_void_p(grad_shard.data_ptr()),
                                            which, self._stream()))

    big_tensor_id = T_DEC_V3      # the decoder's output layer: the one tensor worth sharding across ranks

    def big_grad(self):
        """(tensor id, padded gradient view [rows, ld], padded parameter view [rows, ld]) of the one
        tensor worth sharding across data-parallel ranks: the decoder's output layer."""
        return T_DEC_V3, self.tensor(T_GRAD + T_DEC_V3, padded=True), self.tensor(T_DEC_V3, padded=True)

    def _w1_layout(self, cap):
        """(rows, header words, total floats) of a packet with room for `cap` rows (None = the model's w1_cap)."""
        if self._w1_packet is None:
            hw, tot = C.c_int64(), C.c_int64()
            _check(self.lib.aae_w1_packet_floats(self.handle, self.w1_cap, C.byref(hw), C.byref(tot)))
            assert hw.value == self._w1_hdr
            self._w1_small = tot.value - hw.value - self.w1_cap * self.h
            self._w1_packet = torch.zeros(tot.value, dtype=torch.float32, device=self.device)
        cap = self.w1_cap if cap is None else max(1, min(int(cap), self.w1_cap))
        hdr = (1 + cap + 3) & ~3
        return cap, hdr, hdr + cap * self.h + self._w1_small

    def w1_export(self, cap=None):
        """Pack this rank's first-layer gradient rows: one flat float32 tensor = int32 header (count, item ids) +
        rows [cap, n_hidden] + the encoder's small-layer gradients.  cap: an upper bound on the distinct items of
        this exchange that every rank agrees on (e.g. the largest entry count of any rank's share of the batch);
        the default is the model-wide worst case w1_cap - the packet is what the all-gather moves."""
        cap, hdr, total = self._w1_layout(cap)
        pk = self._w1_packet[:total]
        with self._on_device():
            _check(self.lib.aae_w1_export(self.handle, C.c_void_p(pk.data_ptr()), C.c_void_p(pk.data_ptr() + 4 * hdr), cap,
                                          self._stream()))
        return pk

    def w1_import(self, packets, n_peers, which, cap=None, stride_floats=None):
        """Sum the peers' packed rows (flat tensor of n_peers packets of w1_export(cap)) and run optimiser `which`.
        stride_floats: distance between consecutive peers' packets when they carry riders behind them."""
        cap, hdr, total = self._w1_layout(cap)
        stride = total if stride_floats is None else int(stride_floats)
        assert stride >= total and packets.numel() >= (n_peers - 1) * stride + total
        self._keep.append(packets)
        with self._on_device():
            _check(self.lib.aae_w1_import(self.handle, C.c_void_p(packets.data_ptr()),
                                          C.c_void_p(packets.data_ptr() + 4 * hdr), cap, n_peers, 4 * stride, which,
                                          self._stream()))

    def set_rng_rows(self, row_offset, global_rows):
        """Device RNG under data parallelism: this rank's rows are [row_offset, row_offset + n_rows) of a global batch
        of global_rows; with one seed on every rank the ranks draw what a single process draws for the whole batch."""
        _check(self.lib.aae_set_rng_rows(self.handle, int(row_offset), int(global_rows)))

    def set_grad_scale(self, scale):
        _check(self.lib.aae_set_grad_scale(self.handle, float(scale)))

    def profile_enable(self, on=True, kernels=None):
        """kernels: iterable of kernel ids to time (None = all)."""
        if on and kernels is not None:
            on = sum(1 << (k + 1) for k in kernels)
        _check(self.lib.aae_profile_enable(self.handle, int(on)))

    def profile_read(self, kernel_id):
        """(total_ms, launches) of the hipEvent pairs recorded around `kernel_id` since the last read."""
        ms, n = C.c_double(), C.c_int64()
        _check(self.lib.aae_profile_read(self.handle, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def losses(self):
        out = (C.c_float * 3)()
        with self._on_device():
            _check(self.lib.aae_read_losses(self.handle, C.byref(out), self._stream()))
        return float(out[0]), float(out[1]), float(out[2])

    # ---- predict ------------------------------------------------------------------------
    def _out_buffer(self, n_rows):
        ld = (self.N + 3) & ~3
        return torch.empty(n_rows, ld, dtype=torch.float32, device=self.device)

    def predict(self, csr, row_start, n_rows, cond=None):
        b = self._batch(csr, row_start, n_rows)
        out = self._out_buffer(n_rows)
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous()
        with self._on_device():
            _check(self.lib.aae_predict(self.handle, C.byref(b), _ptr(cond), _ptr(out), out.shape[1], self._stream()))
        return out[:, :self.N]

    def rank_max_rows(self, k=10):
        """Rows one predict_topk / decode_topk call may rank (aae_rank_max_rows): far more than max_batch where the fused
        predict -> rank kernels apply, max_batch otherwise."""
        key = int(k)
        if key not in self._rank_rows:
            out = C.c_int32()
            _check(self.lib.aae_rank_max_rows(self.handle, key, C.byref(out)))
            self._rank_rows[key] = int(out.value)
        return self._rank_rows[key]

    def predict_topk(self, csr, row_start, n_rows, k, cond=None, exclude_known=True):
        """(ids int32 [n_rows, k], scaled scores float32 [n_rows, k]) - device tensors.  n_rows <= rank_max_rows(k).
        Ties: the reference leaves items of equal fp32 score in np.argpartition's order (evaluation.py:20-58).  Calls within
        the fused path's row limit order such items by their LOGIT (saturated sigmoids included), the dense fallback by the
        smaller item id: the k scores are the same either way, the items named may differ exactly where scores tie
        (tests/test_rank_gpu.py::test_saturated_scores_tie_and_both_rank_paths_return_a_valid_top_k)."""
        b = self._batch(csr, row_start, n_rows, bounded=n_rows <= self.max_batch)     # (beyond max_batch: the fused rank path, which has no per-batch lists)
        idx = torch.empty(n_rows, k, dtype=torch.int32, device=self.device)
        val = torch.empty(n_rows, k, dtype=torch.float32, device=self.device)
        if cond is not None:
            cond = upload(cond, self.device, torch.float32).contiguous()
        with self._on_device():
            _check(self.lib.aae_predict_topk(self.handle, C.byref(b), _ptr(cond), int(k), int(bool(exclude_known)),
                                             _ptr(idx), _ptr(val), self._stream()))
        return idx, val

    def decode_topk(self, zc, csr, row_start, k, exclude_known=True):
        """Top-k of decode(zc) for the input rows csr[row_start : row_start + len(zc)] (aae_decode_topk)."""
        zc = zc.detach().to(self.device, torch.float32).contiguous()
        n_rows = zc.shape[0]
        b = self._batch(csr, row_start, n_rows, bounded=n_rows <= self.max_batch)
        idx = torch.empty(n_rows, k, dtype=torch.int32, device=self.device)
        val = torch.empty(n_rows, k, dtype=torch.float32, device=self.device)
        with self._on_device():
            _check(self.lib.aae_decode_topk(self.handle, _ptr(zc), zc.shape[1], C.byref(b), int(k), int(bool(exclude_known)),
                                            _ptr(idx), _ptr(val), self._stream()))
        return idx, val

    def encode(self, csr, row_start, n_rows):
        b = self._batch(csr, row_start, n_rows)
        z = torch.empty(n_rows, self.c, dtype=torch.float32, device=self.device)
        with self._on_device():
            _check(self.lib.aae_encode(self.handle, C.byref(b), _ptr(z), self._stream()))
        return z

    def decode(self, zc):
        zc = zc.detach().to(self.device, torch.float32).contiguous()
        out = self._out_buffer(zc.shape[0])
        with self._on_device():
            _check(self.lib.aae_decode(self.handle, _ptr(zc), zc.shape[1], zc.shape[0], _ptr(out), out.shape[1],
                                       self._stream()))
        return out[:, :self.N]
