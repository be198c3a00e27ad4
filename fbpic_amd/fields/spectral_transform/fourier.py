"""Batched FFT along z.  Replaces the cuFFT path of
fbpic/fields/spectral_transform/fourier.py:27-168: no transpose copies, the 1/Nz of the
backward transform is part of the kernel / plan.  Power-of-two Nz in [64, 4096] run the
hand-written LDS Stockham kernel (csrc/zfft.hip), as do 9 * 2^k; other lengths go through
rocFFT (csrc/fft.hip), and the ones rocFFT refuses through the library's generic
pass-per-launch FFT (fb_fft_generic)."""
from ... import _capi
import ctypes

_PLANS = {}
# set to False to force the rocFFT path (A/B measurements: tools/kbench.py --rocfft)
USE_ZFFT = True


def get_plan(Nz, ncols, in_stride, out_stride, inplace=False):
    """rocFFT plan for a (Nz, ncols) strided view (cached for the life of the process).
    Returns None when rocFFT refuses the strided layout for this length."""
    key = (Nz, ncols, in_stride, out_stride, bool(inplace))
    if key not in _PLANS:
        h = ctypes.c_void_p()
        rc = _capi.lib().fb_fft_plan_create(Nz, ncols, in_stride, out_stride, int(inplace),
                                            ctypes.byref(h))
        _PLANS[key] = h if rc == 0 else None
    return _PLANS[key]


def generic_scratch(Nz, ncols, device):
    """Scratch slab of the pass-per-launch FFT (one per length and column count)."""
    t = _capi.torch()
    key = ('G', Nz, ncols, str(device))
    if key not in _PLANS:
        _PLANS[key] = t.empty((Nz, ncols + 8), dtype=t.complex128, device=device)
    return _PLANS[key]


def _generic_exec(src, dst, direction, Nz, ncols):
    """Fallback for lengths rocFFT refuses (e.g. 4416 = 2^6 * 3 * 23): the library's own
    pass-per-launch FFT, ping-pong with a scratch slab; lengths with a prime factor > 31
    (e.g. 4288 = 2^6 * 67) go through Bluestein's algorithm on top of it."""
    t = _capi.torch()
    if not _capi.lib().fb_fft_generic_supported(Nz):
        _bluestein_exec(src, dst, direction, Nz, ncols)
        return
    scratch = generic_scratch(Nz, ncols, src.device)
    rc = _capi.lib().fb_fft_generic(Nz, ncols, src.data_ptr(), src.stride(0), dst.data_ptr(),
                                    dst.stride(0), scratch.data_ptr(), scratch.stride(0),
                                    direction, _capi.stream())
    _capi.check(rc, 'fb_fft_generic')


def _smooth_length(n):
    """Smallest 2^a 3^b 5^c >= n."""
    best = 1 << (n - 1).bit_length()
    p5 = 1
    while p5 < best:
        p35 = p5
        while p35 < best:
            m = p35
            while m < n:
                m *= 2
            best = min(best, m)
            p35 *= 3
        p5 *= 5
    return best


def _bluestein_exec(src, dst, direction, Nz, ncols):
    """DFT of arbitrary length as a circular convolution of smooth length M >= 2 Nz - 1:
    X[k] = conj(b[k]) sum_n x[n] conj(b[n]) b[k - n], b[n] = exp(i pi n^2 / Nz).  The backward
    transform is conj(forward(conj(x))) / Nz.  Everything stays on the device."""
    import numpy as np
    t = _capi.torch()
    M = _smooth_length(2 * Nz - 1)
    key = ('B', Nz, ncols)
    if key not in _PLANS:
        n = np.arange(Nz, dtype=np.int64)
        b = np.exp(1j * np.pi * ((n * n) % (2 * Nz)) / Nz)        # n^2 mod 2 Nz keeps the phase exact
        ext = np.zeros(M, dtype=np.complex128)
        ext[:Nz] = b
        ext[M - Nz + 1:] = b[1:][::-1]
        dev = src.device
        _PLANS[key] = (t.as_tensor(np.conj(b), device=dev), t.as_tensor(np.fft.fft(ext), device=dev),
                       t.zeros((M, ncols + 8), dtype=t.complex128, device=dev),
                       t.empty((M, ncols + 8), dtype=t.complex128, device=dev))
    bc, Bf, work, scratch = _PLANS[key]
    sv = t.as_strided(src, (Nz, ncols), (src.stride(0), 1))
    dv = t.as_strided(dst, (Nz, ncols), (dst.stride(0), 1))
    a = work[:, :ncols]
    a[Nz:] = 0.
    a[:Nz] = (sv if direction < 0 else sv.conj()) * bc[:, None]
    lib, st = _capi.lib(), _capi.stream()
    for sign in (-1, +1):
        rc = lib.fb_fft_generic(M, ncols, work.data_ptr(), work.stride(0), work.data_ptr(),
                                work.stride(0), scratch.data_ptr(), scratch.stride(0), sign, st)
        _capi.check(rc, 'fb_fft_generic')
        if sign < 0:
            a *= Bf[:, None]
    out = a[:Nz] * bc[:, None]
    dv.copy_(out if direction < 0 else out.conj() / Nz)


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
    if plan is None:
        _generic_exec(src, dst, direction, Nz, ncols)
        return
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
