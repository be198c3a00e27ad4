"""Batched FFT along z.  Replaces the cuFFT path of
fbpic/fields/spectral_transform/fourier.py:27-168: no transpose copies, the 1/Nz of the
backward transform is part of the kernel / plan.  Power-of-two Nz in [64, 4096] run the
hand-written LDS Stockham kernel (csrc/zfft.hip); every other length goes through rocFFT
(csrc/fft.hip)."""
from ... import _capi
import ctypes

_PLANS = {}
# set to False to force the rocFFT path (A/B measurements: tools/kbench.py --rocfft)
USE_ZFFT = True


def get_plan(Nz, ncols, in_stride, out_stride, inplace=False):
    """rocFFT plan for a (Nz, ncols) strided view (cached for the life of the process)."""
    key = (Nz, ncols, in_stride, out_stride, bool(inplace))
    if key not in _PLANS:
        h = ctypes.c_void_p()
        rc = _capi.lib().fb_fft_plan_create(Nz, ncols, in_stride, out_stride, int(inplace),
                                            ctypes.byref(h))
        _capi.check(rc, 'fb_fft_plan_create')
        _PLANS[key] = h
    return _PLANS[key]


def fft_exec(src, dst, direction, ncols=None):
    """Transform the (Nz, ncols) view starting at src into dst (device tensors whose first
    stride is the row stride).  direction = -1 forward, +1 backward (scaled by 1/Nz)."""
    Nz = src.shape[0]
    if ncols is None:
        ncols = src.shape[1]
    inplace = src.data_ptr() == dst.data_ptr()
    if USE_ZFFT and _capi.lib().fb_zfft_supported(Nz):
        rc = _capi.lib().fb_zfft(Nz, ncols, src.data_ptr(), src.stride(0), dst.data_ptr(),
                                 dst.stride(0), direction, _capi.stream())
        _capi.check(rc, 'fb_zfft')
        return
    plan = get_plan(Nz, ncols, src.stride(0), dst.stride(0), inplace)
    rc = _capi.lib().fb_fft_exec(plan, direction, src.data_ptr(), dst.data_ptr(), _capi.stream())
    _capi.check(rc, 'fb_fft_exec')


class FFT(object):
    """Same interface as the reference's FFT object (fourier.py:27-168)."""

    def __init__(self, Nr, Nz, use_cuda=True, nthreads=None):
        self.Nr, self.Nz, self.use_cuda = Nr, Nz, use_cuda

    def transform(self, array_in, array_out):
        fft_exec(array_in, array_out, -1)

    def inverse_transform(self, array_in, array_out):
        fft_exec(array_in, array_out, +1)
