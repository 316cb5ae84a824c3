"""ctypes binding of libttl_hip.so (C ABI: include/ttl_hip.h).

There is no CPU fallback: if the shared library is missing or does not load,
``load()`` raises.  Build it with ``python -m tracktolearn_amd.csrc.build``
(or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME
# libamdhip64.so.7).  It must be in the process BEFORE libttl_hip.so is
# dlopen-ed so that the dynamic loader binds our NEEDED libamdhip64.so.7 to
# that same copy; loaded the other way round, /opt/rocm's runtime comes in
# first, torch then loads its own as a second runtime, and our kernels run on
# a runtime that has no device ("no ROCm-capable device is detected").
import torch  # noqa: F401  (keep above the CDLL below)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libttl_hip.so')

ABI_VERSION = 11
SH_LINEAR, SH_BRICK4 = 0, 1
MODE_F32 = 0
MODE_F64DIR = 1
MODE_F32NORM = 2
ORDER_ACTIVE = 0
ORDER_PARTITION = 1
ORDER_BY_POSITION = C.c_void_p(1)      # TTL_ORDER_BY_POSITION (ttl_env_reset)
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = -1, -2, -3, -4


class TTLError(RuntimeError):
    pass


class EnvDesc(C.Structure):
    """struct ttl_env_desc (include/ttl_hip.h) -- field order is the ABI."""
    _fields_ = [
        ('abi_version', C.c_uint32),
        ('mode', C.c_int32),
        ('sh_dim', C.c_int32 * 3),
        ('n_coef', C.c_int32),
        ('coef_pitch', C.c_int32),
        ('sh_packed', C.c_void_p),
        ('sh_coord_shift', C.c_float),
        ('sh_layout', C.c_int32),
        ('mask_dim', C.c_int32 * 3),
        ('mask_coef', C.c_void_p),
        ('mask_threshold', C.c_double),
        ('mask_classes', C.c_void_p),
        ('peaks_dim', C.c_int32 * 3),
        ('peaks', C.c_void_p),
        ('compute_reward', C.c_int32),
        ('alignment_weighting', C.c_double),
        ('n_dirs', C.c_int32),
        ('max_nb_steps', C.c_int32),
        ('step_size_vox', C.c_double),
        ('neigh_radius_vox', C.c_float),
        ('curvature_enabled', C.c_int32),
        ('curv_dot_max', C.c_float),
        ('n_max', C.c_int32),
        ('streamlines', C.c_void_p),
        ('flags', C.c_void_p),
        ('lengths', C.c_void_p),
        ('dones', C.c_void_p),
        ('idx_a', C.c_void_p),
        ('idx_b', C.c_void_p),
        ('workspace', C.c_void_p),
        ('workspace_bytes', C.c_size_t),
    ]


class ColsumSeg(C.Structure):
    """struct ttl_colsum_seg (include/ttl_learner.h)."""
    _fields_ = [
        ('part', C.c_void_p),
        ('ld', C.c_int64),
        ('n_part', C.c_int32),
        ('n', C.c_int32),
        ('out', C.c_void_p),
        ('scale', C.c_float),
        ('accumulate', C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/ttl_hip.h and
# include/ttl_learner.h declare
SYMBOLS = {
    'ttl_env_workspace_bytes': (C.c_size_t, [C.c_int32]),
    'ttl_sh_volume_records': (C.c_int64, [C.POINTER(C.c_int32), C.c_int32]),
    'ttl_volume_alloc': (C.c_int, [C.c_int32, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int32)]),
    'ttl_volume_free': (C.c_int, [C.c_void_p]),
    'ttl_pack_sh_volume': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32),
                                      C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'ttl_mask_classes': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_double,
                                   C.c_void_p, C.c_void_p]),
    'ttl_env_create': (C.c_int, [C.POINTER(EnvDesc), C.POINTER(C.c_void_p)]),
    'ttl_env_destroy': (None, [C.c_void_p]),
    'ttl_env_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p]),
    'ttl_env_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                               C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttl_env_step_begin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    'ttl_env_step_end': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p]),
    'ttl_env_wait_counts': (C.c_int, [C.c_void_p]),
    'ttl_env_stopped': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    'ttl_oracle_segments': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'ttl_oracle_bonus': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double,
                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttl_env_harvest_wait': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.POINTER(C.c_int32)]),
    'ttl_env_harvest': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int64, C.c_void_p]),
    'ttl_env_freerun_begin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttl_env_freerun_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttl_env_freerun_scripted_actions': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                                   C.c_int32, C.c_int32, C.c_uint32,
                                                   C.c_float, C.c_void_p, C.c_void_p]),
    'ttl_env_freerun_end': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.c_void_p]),
    'ttl_env_stopping_flags': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_void_p]),
    'ttl_env_set_processing_order': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32,
                                               C.c_void_p]),
    'ttl_env_refresh_processing_order': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ttl_env_view': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p),
                               C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    'ttl_env_profile_begin': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'ttl_env_profile_end': (C.c_int, [C.c_void_p, C.POINTER(C.c_double),
                                      C.POINTER(C.c_int32)]),
    'ttl_scripted_actions': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_void_p, C.c_int32, C.c_uint32,
                                       C.c_uint32, C.c_float, C.c_void_p,
                                       C.c_void_p]),
    'ttl_peaks_from_sh': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_float, C.c_float, C.c_float,
                                    C.c_int32, C.c_void_p, C.c_void_p]),
    'ttl_resample_streamlines': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_void_p]),
    'ttl_oracle_net_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_void_p]),
    'ttl_pack_streamlines': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_void_p]),
    # ---- include/ttl_learner.h
    'ttl_thin_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttl_sac_losses': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                 C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    'ttl_thin_backward': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                    C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    'ttl_relu_backward_bias': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_int64, C.c_void_p]),
    'ttl_colsum_finalize': (C.c_int, [C.POINTER(ColsumSeg), C.c_int32, C.c_void_p]),
    'ttl_sac_actor_head_backward': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                              C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                              C.c_void_p]),
    'ttl_td3_losses': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                 C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_double,
                                 C.c_double, C.c_double, C.c_void_p]),
    'ttl_polyak_average': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                     C.c_void_p]),
    'ttl_adam_polyak': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int64, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_void_p]),
    'ttl_sac_alpha_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_float, C.c_void_p, C.c_double, C.c_double,
                                     C.c_double, C.c_void_p]),
    'ttl_build_learner_inputs': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    'ttl_replay_add': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttl_replay_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                    C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttl_last_error': (C.c_char_p, []),
    'ttl_abi_version': (C.c_uint32, []),
    'ttl_env_desc_size': (C.c_size_t, []),
}

_lib = None


def load():
    """Load libttl_hip.so once; raise if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TTLError(
            f'{LIB_PATH} not found: the HIP extension is required (build it '
            'with `python -m tracktolearn_amd.csrc.build`); there is no CPU '
            'fallback')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ttl_abi_version() != ABI_VERSION:
        raise TTLError('libttl_hip.so ABI version mismatch')
    if lib.ttl_env_desc_size() != C.sizeof(EnvDesc):
        raise TTLError('ttl_env_desc layout mismatch between libttl_hip.so '
                       f'({lib.ttl_env_desc_size()} B) and the ctypes mirror '
                       f'({C.sizeof(EnvDesc)} B)')
    _lib = lib
    return lib


class DeviceVolume:
    """Device memory from ``ttl_volume_alloc`` (on request physically contiguous
    when the driver grants it), exposed through ``__cuda_array_interface__`` so that
    ``torch.as_tensor(vol, device=...)`` wraps it without a copy; the tensor
    keeps this object alive, ``__del__`` returns the memory."""

    def __init__(self, device_index, nbytes, try_contiguous=False):
        lib = load()
        ptr, contiguous = C.c_void_p(), C.c_int32()
        check(lib.ttl_volume_alloc(int(device_index), int(nbytes), int(try_contiguous),
                                   C.byref(ptr), C.byref(contiguous)), 'ttl_volume_alloc')
        self.ptr, self.nbytes, self.contiguous = ptr.value, int(nbytes), int(contiguous.value)
        self.__cuda_array_interface__ = {'shape': (self.nbytes,), 'typestr': '|u1',
                                         'data': (self.ptr, False), 'version': 2}

    def __del__(self):
        ptr, self.ptr = getattr(self, 'ptr', None), None
        if ptr and _lib is not None:
            _lib.ttl_volume_free(ptr)


def check(code, what=''):
    if code != 0:
        msg = load().ttl_last_error().decode('utf-8', 'replace')
        raise TTLError(f'{what} failed ({code}): {msg}')
