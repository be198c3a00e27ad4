"""Interpolation grid <-> spectral grid for one azimuthal mode: FFT(z) o DHT(r).
Per-array interface of fbpic/fields/spectral_transform/spectral_transformer.py:21-223.
(`Fields` uses a batched path over all modes / components instead, see fields.py.)"""
from ... import _capi
from .hankel import DHT
from .fourier import FFT


class SpectralTransformer(object):
    def __init__(self, Nz, Nr, m, rmax, use_cuda=True):
        self.use_cuda = use_cuda
        self.Nz, self.Nr, self.m = Nz, Nr, m
        self.dht0 = DHT(m, m, Nr, Nz, rmax, use_cuda=use_cuda)
        self.dhtp = DHT(m + 1, m, Nr, Nz, rmax, use_cuda=use_cuda)
        self.dhtm = DHT(m - 1, m, Nr, Nz, rmax, use_cuda=use_cuda)
        self.fft = FFT(Nr, Nz, use_cuda=use_cuda)
        self.spect_buffer_r = None
        self.spect_buffer_t = None

    def _buffers(self):
        if self.spect_buffer_r is None:
            t = _capi.torch()
            dev = _capi.require_device()
            self.spect_buffer_r = t.zeros((self.Nz, self.Nr), dtype=t.complex128, device=dev)
            self.spect_buffer_t = t.zeros((self.Nz, self.Nr), dtype=t.complex128, device=dev)
            self.spect_buffer_p = self.spect_buffer_r
            self.spect_buffer_m = self.spect_buffer_t
        return self.spect_buffer_r, self.spect_buffer_t

    def spect2interp_scal(self, spect_array, interp_array):
        br, _ = self._buffers()
        self.dht0.inverse_transform(spect_array, br)
        self.fft.inverse_transform(br, interp_array)

    def spect2interp_vect(self, spect_array_p, spect_array_m, interp_array_r, interp_array_t):
        br, bt = self._buffers()
        self.dhtp.inverse_transform(spect_array_p, br)
        self.dhtm.inverse_transform(spect_array_m, bt)
        lib = _capi.lib()
        pa = _capi.ptr_array
        _capi.check(lib.fb_pm_to_rt(1, pa([br]), pa([bt]), pa([br]), pa([bt]), br.stride(0),
                                    self.Nz, self.Nr, _capi.stream()), 'fb_pm_to_rt')
        self.fft.inverse_transform(br, interp_array_r)
        self.fft.inverse_transform(bt, interp_array_t)

    def interp2spect_scal(self, interp_array, spect_array):
        br, _ = self._buffers()
        self.fft.transform(interp_array, br)
        self.dht0.transform(br, spect_array)

    def interp2spect_vect(self, interp_array_r, interp_array_t, spect_array_p, spect_array_m):
        br, bt = self._buffers()
        self.fft.transform(interp_array_r, br)
        self.fft.transform(interp_array_t, bt)
        lib = _capi.lib()
        pa = _capi.ptr_array
        _capi.check(lib.fb_rt_to_pm(1, pa([br]), pa([bt]), pa([br]), pa([bt]), br.stride(0),
                                    self.Nz, self.Nr, _capi.stream()), 'fb_rt_to_pm')
        self.dhtp.transform(br, spect_array_p)
        self.dhtm.transform(bt, spect_array_m)
