from .fields import Fields
from .interpolation_grid import InterpolationGrid
from .spectral_grid import SpectralGrid
from .psatd_coefs import PsatdCoeffs
from .smoothing import BinomialSmoother
__all__ = ['Fields', 'InterpolationGrid', 'SpectralGrid', 'PsatdCoeffs', 'BinomialSmoother']
