"""ctypes binding of libfbpic_amd.so (include/fbpic_amd.h) + device-array plumbing.

The HIP library is the only execution path of this package: if it is missing, or no
MI355X is visible, every compute entry point raises (there is no CPU fallback).
PyTorch-ROCm is used purely for device memory and streams: a device array is a
`torch.Tensor` on `cuda`, kernels receive `tensor.data_ptr()` and the current stream.
"""
import ctypes
import os
from ctypes import c_double as D, c_int as I, c_long as L, c_void_p as P, c_size_t as Z

_HERE = os.path.dirname(os.path.abspath(__file__))
# FBPIC_AMD_LIB: developer override (A/B timing of two builds of the library)
LIB_PATH = os.environ.get('FBPIC_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libfbpic_amd.so')
_lib = None

_PP = ctypes.POINTER(P)   # host array of device pointers

_SIGNATURES = {
    'fb_abi_version': (I, []),
    'fb_last_error': (ctypes.c_char_p, []),
    'fb_build_info': (ctypes.c_char_p, []),
    'fb_last_error_string': (ctypes.c_char_p, []),
    'fb_malloc': (I, [Z, ctypes.POINTER(P)]),
    'fb_free': (I, [P]),
    'fb_h2d': (I, [P, P, Z, P]),
    'fb_d2h': (I, [P, P, Z, P]),
    'fb_set_device': (I, [I]),
    'fb_sync': (I, [P]),
    'fb_comm_unique_id': (I, [P]),
    'fb_comm_init': (I, [P, I, I, ctypes.POINTER(P)]),
    'fb_comm_destroy': (I, [P]),
    'fb_exchange': (I, [P, I, I, P, Z, P, Z, P, Z, P, Z, P]),
    'fb_push_x': (I, [L, P, P, P, P, P, P, P, D, D, D, D, D, P]),
    'fb_push_p': (I, [L, P, P, P, P, P, P, P, P, P, P, D, D, D, D, P]),
    'fb_shift_periodic': (I, [L, P, D, D, P]),
    'fb_gather': (I, [I, I, L, P, P, P, D, D, D, I, D, D, I, _PP, L, P, P, P, P, P, P, P]),
    'fb_gather_push': (I, [I, I, L, P, P, P, P, P, P, P, D, D, D, I, D, D, I, _PP, L,
                           P, P, P, P, P, P, D, D, D, D, D, D, D, P]),
    'fb_gather_push_rank_next': (I, [I, I, L, P, P, P, P, P, P, P, D, D, D, I, D, D, I, _PP, L,
                                     P, P, P, P, P, P, D, D, D, D, D, D, D, D, D, D, D, I, P, Z, I, P]),
    'fb_gather_push_rank_next_range': (I, [I, I, L, P, P, P, P, P, P, P, D, D, D, I, D, D, I, _PP, L,
                                           P, P, P, P, P, P, D, D, D, D, D, D, D, D, D, D, D, I, P, Z, I,
                                           P, P, I, P]),
    'fb_cell_index': (I, [L, P, P, P, D, D, I, D, D, I, P, P, P]),
    'fb_sort_workspace_bytes': (Z, [L, I]),
    'fb_sort_by_cell': (I, [L, I, P, P, P, P, ctypes.POINTER(I), P, P, Z, P]),
    'fb_bin_sort_workspace_bytes': (Z, [L, I]),
    'fb_bin_sort_particles': (I, [L, I, P, P, P, D, D, I, D, D, I, I, _PP, _PP, P, P, P, P, Z, P]),
    'fb_push_x_bin_sort_particles': (I, [L, I, P, P, P, P, P, P, P, D, D, D, D, D, D, D, I, D, D, I, I, _PP,
                                         _PP, P, P, P, P, Z, I, P]),
    'fb_deposit_J_rank_next': (I, [I, I, L, P, P, P, P, D, P, P, P, P, D, D, D, I, D, D, I, _PP, L, L, P, P, P,
                                   D, D, D, D, I, P, Z, I, P]),
    'fb_push_x_sort_deposit_rho': (I, [L, I, P, P, P, P, P, P, P, D, D, D, D, D, D, D, I, D, D, I, I, _PP,
                                       _PP, P, P, P, P, Z, I, I, I, D, _PP, L, L, P, P, P]),
    'fb_push_x_sort_deposit_J_rho': (I, [L, I, P, P, P, P, P, P, P, D, D, D, D, D, D, D, I, D, D, I, I, _PP,
                                         _PP, P, P, P, P, Z, I, I, I, D, D, _PP, L, L, _PP, L, L, P, P, I, P]),
    'fb_gather_push_deposit_supported': (I, [I, I]),
    'fb_gather_push_deposit_J_rho': (I, [I, I, L, P, P, P, P, P, P, P, P, P, D, D, D, I, D, D, I, _PP, L,
                                         P, P, P, P, P, P, D, D, D, D, D, D, D, _PP, L, L, _PP, L, L,
                                         P, P, P, I, P]),
    'fb_gather_push_rank_next_home': (I, [I, I, L, P, P, P, P, P, P, P, P, D, D, D, I, D, D, I, _PP, L,
                                          P, P, P, P, P, P, D, D, D, D, D, D, D, I, P, Z, I, I, P]),
    'fb_permute': (I, [L, P, I, _PP, _PP, P]),
    'fb_handover_pack': (I, [L, P, I, _PP, P, L, P]),
    'fb_handover_move': (I, [L, P, P, I, _PP, P]),
    'fb_handover_append': (I, [L, L, I, _PP, P, L, P]),
    'fb_handover_select_pack': (I, [L, P, P, L, L, L, L, D, D, I, _PP, L, L, L, P, P, P, P, P, P]),
    'fb_handover_recv_counts': (I, [P, P, P, P]),
    'fb_handover_workspace_bytes': (Z, [L]),
    'fb_handover_compact': (I, [L, L, P, L, P, I, _PP, P, Z, P]),
    'fb_handover_append_shift': (I, [L, L, I, _PP, P, L, I, D, P]),
    'fb_deposit_rho': (I, [I, I, L, P, P, P, P, D, D, D, I, D, D, I, _PP, L, L, P, P, P, P, P]),
    'fb_deposit_J': (I, [I, I, L, P, P, P, P, D, P, P, P, P, D, D, D, I, D, D, I, _PP, L, L,
                         P, P, P, P, P]),
    'fb_erase': (I, [I, _PP, L, I, I, P]),
    'fb_divide_by_volume': (I, [I, _PP, L, P, I, I, P]),
    'fb_filter': (I, [I, _PP, L, P, P, I, I, P]),
    'fb_correct_currents_curlfree_standard': (I, [P, P, P, P, P, L, P, P, P, D, I, I, P]),
    'fb_push_eb_standard': (I, [P] * 11 + [L] + [P] * 7 + [D, I, D, D, D, I, I, P]),
    'fb_correct_currents_curlfree_comoving': (I, [P, P, P, P, P, L, P, P, P, P, P, P, I, I, P]),
    'fb_correct_currents_crossdeposition_standard': (I, [P] * 7 + [L, P, P, D, I, I, P]),
    'fb_correct_currents_crossdeposition_comoving': (I, [P] * 7 + [L, P, P, P, P, P, I, I, P]),
    'fb_push_eb_comoving': (I, [P, P, P, P, P, P, P, P, P, P, P, L, P, P, P, P, P, P, P, P, P, P, D, D, I, D, D, D,
                                I, I, P]),
    'fb_push_rho': (I, [P, P, L, I, I, P]),
    'fb_rt_to_pm': (I, [I, _PP, _PP, _PP, _PP, L, I, I, P]),
    'fb_pm_to_rt': (I, [I, _PP, _PP, _PP, _PP, L, I, I, P]),
    'fb_guard_buffers': (I, [I, P, L, L, I, I, I, P, P, P]),
    'fb_damp_rows': (I, [P, L, L, P, I, P, I, I, P]),
    'fb_shift_spect': (I, [I, _PP, L, P, I, I, I, P]),
    'fb_scale': (I, [I, _PP, L, D, I, I, P]),
    'fb_fft_plan_create': (I, [I, L, L, L, I, _PP]),
    'fb_fft_exec': (I, [P, I, P, P, P]),
    'fb_fft_plan_destroy': (I, [P]),
    'fb_zfft_supported': (I, [I]),
    'fb_zfft': (I, [I, L, P, L, P, L, I, P]),
    'fb_zfft_pm_to_rt': (I, [I, L, P, L, P, L, I, P]),
    'fb_zfft_from_records': (I, [I, I, I, P, L, I, P, L, P]),
    'fb_zfft_from_records_consume': (I, [I, I, I, P, L, I, P, L, P]),
    'fb_fft_generic_supported': (I, [I]),
    'fb_fft_generic_from_records_supported': (I, [I]),
    'fb_fft_generic_from_records_consume': (I, [I, I, I, P, L, I, P, L, P, L, P]),
    'fb_fft_generic': (I, [I, L, P, L, P, L, P, L, I, P]),
    'fb_hankel': (I, [I, _PP, L, _PP, L, _PP, D, I, I, P]),
    'fb_hankel_scaled': (I, [I, _PP, L, _PP, L, _PP, _PP, _PP, _PP, D, I, I, P]),
    'fb_hankel_pm_to_rt': (I, [I, _PP, _PP, L, _PP, _PP, L, _PP, _PP, D, I, I, P]),
    'fb_hankel_rt_to_pm_scaled': (I, [I, _PP, _PP, P, L, _PP, L, _PP, _PP, _PP, _PP, D, I, I, P]),
    'fb_spect_cycle_supported': (I, [I, I]),
    'fb_spect_cycle_standard': (I, [I, _PP, L, _PP, _PP, _PP, _PP, _PP, _PP, L, _PP, D, I, I, D, D, D,
                                    _PP, L, I, I, P]),
    'fb_psatd_step_standard': (I, [I, _PP, L, _PP, D, I, I, D, D, D, I, I, P]),
    'fb_psatd_step_standard_shift': (I, [I, _PP, L, _PP, D, I, I, D, D, D, I, I, P, I, P]),
}

EXPORTS = tuple(_SIGNATURES)
# FB_ABI_VERSION of include/fbpic_amd.h the signatures above were written against
ABI_VERSION = 7


class BackendError(RuntimeError):
    pass


def lib():
    """Load libfbpic_amd.so (raises BackendError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BackendError(
                'libfbpic_amd.so not found at %s: build it with '
                '`make -C fbpic_amd/csrc` (or __graft_entry__.build()). '
                'fbpic_amd has no CPU fallback.' % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64 / librocfft / libhiprtc (same SONAMEs as
        # /opt/rocm).  Import torch FIRST so that the process holds exactly one HIP runtime:
        # the DT_NEEDED entries of libfbpic_amd.so then resolve to the copies torch loaded.
        torch()
        _l = ctypes.CDLL(LIB_PATH)
        # the argument lists below belong to ONE revision of include/fbpic_amd.h: a library built
        # from another one must not be bound (a stale .so would otherwise fail far from here, with
        # an AttributeError or - worse - shifted arguments)
        try:
            _l.fb_abi_version.restype = I
            have = _l.fb_abi_version()
        except AttributeError:
            have = None
        if have != ABI_VERSION:
            raise BackendError('%s was built for ABI revision %s of include/fbpic_amd.h, this package '
                               'binds revision %d: rebuild it (`make -C fbpic_amd/csrc`).'
                               % (LIB_PATH, have, ABI_VERSION))
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(_l, name)
            f.restype = res
            f.argtypes = args
        _lib = _l
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().fb_last_error().decode(errors='replace')
        raise BackendError('%s failed (code %d): %s' % (what, rc, msg))


_torch = None


def torch():
    global _torch
    if _torch is None:
        import torch as _t
        _torch = _t
    return _torch


def require_device():
    """Return the torch device of the HIP backend or raise: no silent CPU path."""
    t = torch()
    if not t.cuda.is_available():
        raise BackendError('No MI355X/ROCm device visible: fbpic_amd only executes on the GPU '
                           '(there is no CPU fallback).')
    lib()
    return t.device('cuda', t.cuda.current_device())


def stream():
    return torch().cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    """Host array of device pointers (None entries become NULL)."""
    n = len(tensors)
    return (P * n)(*[(t.data_ptr() if t is not None else None) for t in tensors])


def row_stride(t):
    """Row stride (in elements) of a (Nz, Nr) device view with contiguous r."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), 'r must be contiguous'
    return t.stride(0)


def to_device(a, dtype=None):
    """numpy -> device tensor (no-op for device tensors)."""
    t = torch()
    if isinstance(a, t.Tensor):
        return a if a.is_cuda else a.to(require_device())
    import numpy as np
    a = np.ascontiguousarray(a, dtype=dtype)
    return t.from_numpy(a).to(require_device())


def to_host(a):
    t = torch()
    if isinstance(a, t.Tensor):
        return a.detach().cpu().numpy().copy() if not a.is_contiguous() else a.detach().cpu().numpy()
    return a


# ---------------------------------------------------------------------------------
# Optional per-entry-point device timing (HIP events on the launch stream).  Used by
# bench.py to measure the average device duration of each kernel of the PIC cycle.
class _TimedLib(object):
    def __init__(self, real, records):
        self._real = real
        self._records = records

    def __getattr__(self, name):
        f = getattr(self._real, name)
        if not name.startswith('fb_') or name in ('fb_last_error', 'fb_last_error_string', 'fb_build_info', 'fb_abi_version', 'fb_malloc', 'fb_free', 'fb_h2d', 'fb_d2h',
                                                  'fb_sort_workspace_bytes', 'fb_bin_sort_workspace_bytes', 'fb_handover_workspace_bytes', 'fb_fft_plan_create',
                                                  'fb_fft_plan_destroy', 'fb_sync', 'fb_set_device', 'fb_gather_push_deposit_supported', 'fb_spect_cycle_supported', 'fb_comm_unique_id', 'fb_comm_init', 'fb_comm_destroy', 'fb_zfft_supported', 'fb_fft_generic_supported', 'fb_fft_generic_from_records_supported'):
            return f
        t = torch()
        recs = self._records.setdefault(name, [])

        def timed(*args):
            e0 = t.cuda.Event(enable_timing=True)
            e1 = t.cuda.Event(enable_timing=True)
            e0.record()
            rc = f(*args)
            e1.record()
            recs.append((e0, e1, args))
            return rc
        return timed


_timing_records = None
_plain_lib = lib


def enable_timing():
    """Start recording (event pairs are resolved by collect_timing())."""
    global _timing_records, lib
    _timing_records = {}
    real = _plain_lib()
    proxy = _TimedLib(real, _timing_records)
    globals()['lib'] = lambda: proxy


def collect_timing():
    """Stop recording; return {entry point: [(milliseconds, args), ...]}."""
    global _timing_records
    torch().cuda.synchronize()
    out = {}
    for name, recs in (_timing_records or {}).items():
        out[name] = [(e0.elapsed_time(e1), args) for e0, e1, args in recs]
    _timing_records = None
    globals()['lib'] = _plain_lib
    return out
