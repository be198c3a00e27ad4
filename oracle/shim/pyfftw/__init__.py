"""Stand-in for pyfftw backed by numpy.fft (forward unnormalised, backward 1/N),
used ONLY for golden-vector capture in the build container."""
import numpy as np
class FFTW:
    def __init__(self, a, b, axes=(0,), direction='FFTW_FORWARD', threads=1):
        self.a, self.b, self.axes, self.direction = a, b, axes, direction
    def update_arrays(self, new_input_array=None, new_output_array=None):
        self.a, self.b = new_input_array, new_output_array
    def __call__(self):
        f = np.fft.fft if self.direction == 'FFTW_FORWARD' else np.fft.ifft
        self.b[...] = f(self.a, axis=self.axes[0])
