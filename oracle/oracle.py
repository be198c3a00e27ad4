"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module.  It is the *checker* for the HIP product path (fbpic_amd/), never a fallback:
nothing under fbpic_amd/ imports it.

Particle and spectral kernels live in fbpic_oracle.c (restated from the reference's
Numba CPU kernels, file:line cited there).  The two vendor-library operations of the
reference CPU path are restated here with the libraries the reference itself falls back
to: `np.fft` for the z-FFT (fbpic/fields/spectral_transform/fourier.py:98-168, forward
unnormalised / backward 1/Nz) and `np.dot` for the Hankel GEMM
(fbpic/fields/spectral_transform/hankel.py:206-212, 237-243).

Parity PINNED: tests/test_oracle_golden.py checks every entry point against
tests/golden/*.npz, produced by the real reference via oracle/capture_golden.py.
"""
import ctypes
import os
import subprocess
import time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_d = ctypes.c_double
c_i = ctypes.c_int
c_l = ctypes.c_long
c_p = ctypes.c_void_p


def build():
    """Compile libfbpic_oracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, 'libfbpic_oracle.so')
    src = os.path.join(_HERE, 'fbpic_oracle.c')
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'libfbpic_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def build_native():
    """The same source compiled for THIS host (`-O3 -march=native`, contraction still off) into
    oracle/_native/ - the build bench.py's `cpu_baseline` leg times, so that the CPU figure is not
    held back by the portable flags of the checker library (which must run on the build container and on
    the GPU box alike; SURVEY.md 8d asks for -O3 -march=native).  None when the compiler refuses."""
    out = os.path.join(_HERE, '_native')
    so = os.path.join(out, 'libfbpic_oracle_native.so')
    src = os.path.join(_HERE, 'fbpic_oracle.c')
    try:
        os.makedirs(out, exist_ok=True)
        subprocess.check_call(['gcc', '-O3', '-march=native', '-fPIC', '-fopenmp', '-ffp-contract=off',
                               '-Wno-unused-function', '-shared', '-o', so, src, '-lm'],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return so
    except (OSError, subprocess.CalledProcessError):
        return None


def use_library(path):
    """Load `path` instead of the checker build (bench.py: the native build of build_native)."""
    global _LIB
    _LIB = None
    lib(path)


def lib(path=None):
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(path or build())
        _LIB.orc_max_threads.restype = c_i
        from scipy.constants import c, epsilon_0, mu_0
        _LIB.orc_set_constants(c_d(c), c_d(epsilon_0), c_d(mu_0))
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_p)


def _f64(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return _p(a)


def _c128(a):
    assert a.dtype == np.complex128 and a.flags.c_contiguous
    return _p(a)


def max_threads():
    return lib().orc_max_threads()


def set_threads(n):
    lib().orc_set_threads(c_i(n))


# ----------------------------------------------------------------- particles
def push_x(x, y, z, ux, uy, uz, inv_gamma, dt, px=1., py=1., pz=1.):
    lib().orc_push_x(c_l(x.size), _f64(x), _f64(y), _f64(z), _f64(ux), _f64(uy), _f64(uz),
                     _f64(inv_gamma), c_d(dt), c_d(px), c_d(py), c_d(pz))


def push_p(ux, uy, uz, inv_gamma, Ex, Ey, Ez, Bx, By, Bz, q, m, dt):
    lib().orc_push_p(c_l(ux.size), _f64(ux), _f64(uy), _f64(uz), _f64(inv_gamma),
                     _f64(Ex), _f64(Ey), _f64(Ez), _f64(Bx), _f64(By), _f64(Bz),
                     c_d(q), c_d(m), c_d(dt))


def shift_periodic(z, zmin, zmax):
    lib().orc_shift_periodic(c_l(z.size), _f64(z), c_d(zmin), c_d(zmax))


def gather(shape, fused, x, y, z, rmax_gather, invdz, zmin, Nz, invdr, rmin, Nr, grids,
           Ex, Ey, Ez, Bx, By, Bz):
    """grids: list over modes of 6 complex (Nz,Nr) arrays Er,Et,Ez,Br,Bt,Bz."""
    Nm = len(grids)
    flat = [g for gm in grids for g in gm]
    for g in flat:
        assert g.shape == (Nz, Nr)
    ptrs = (c_p * (6 * Nm))(*[_c128(g) for g in flat])
    lib().orc_gather(c_i(1 if shape == 'linear' else 3), c_i(int(fused)), c_i(Nm),
                     c_l(x.size), _f64(x), _f64(y), _f64(z), c_d(rmax_gather),
                     c_d(invdz), c_d(zmin), c_i(Nz), c_d(invdr), c_d(rmin), c_i(Nr), ptrs,
                     _f64(Ex), _f64(Ey), _f64(Ez), _f64(Bx), _f64(By), _f64(Bz))


def cell_index(x, y, z, invdz, zmin, Nz, invdr, rmin, Nr):
    out = np.empty(x.size, dtype=np.int32)
    lib().orc_cell_index(c_l(x.size), _f64(x), _f64(y), _f64(z), c_d(invdz), c_d(zmin),
                         c_i(Nz), c_d(invdr), c_d(rmin), c_i(Nr), _p(out))
    return out


def deposit_rho_global(shape, Nm, x, y, z, w, q, invdz, zmin, Nz, invdr, rmin, Nr,
                       beta0, betah, nthreads=1, glob=None):
    if glob is None:
        glob = np.zeros((nthreads, Nm, Nz + 4, Nr + 4), dtype=np.complex128)
    lib().orc_deposit_rho(c_i(1 if shape == 'linear' else 3), c_i(Nm), c_l(x.size),
                          _f64(x), _f64(y), _f64(z), _f64(w), c_d(q),
                          c_d(invdz), c_d(zmin), c_i(Nz), c_d(invdr), c_d(rmin), c_i(Nr),
                          _c128(glob), c_i(nthreads), _f64(beta0), _f64(betah))
    return glob


def deposit_J_global(shape, Nm, x, y, z, w, q, ux, uy, uz, inv_gamma, invdz, zmin, Nz,
                     invdr, rmin, Nr, beta0, betah, nthreads=1, globs=None):
    if globs is None:
        globs = [np.zeros((nthreads, Nm, Nz + 4, Nr + 4), dtype=np.complex128) for _ in range(3)]
    lib().orc_deposit_J(c_i(1 if shape == 'linear' else 3), c_i(Nm), c_l(x.size),
                        _f64(x), _f64(y), _f64(z), _f64(w), c_d(q),
                        _f64(ux), _f64(uy), _f64(uz), _f64(inv_gamma),
                        c_d(invdz), c_d(zmin), c_i(Nz), c_d(invdr), c_d(rmin), c_i(Nr),
                        _c128(globs[0]), _c128(globs[1]), _c128(globs[2]), c_i(nthreads),
                        _f64(beta0), _f64(betah))
    return globs


def sum_reduce(glob, m, reduced):
    nthreads, Nm, Nz4, Nr4 = glob.shape
    lib().orc_sum_reduce(_c128(glob), c_i(nthreads), c_i(Nm), c_i(Nz4 - 4), c_i(Nr4 - 4),
                         c_i(m), _c128(reduced))


def divide_by_volume(F, invvol):
    Nz, Nr = F.shape
    lib().orc_divide_by_volume(_c128(F), _f64(invvol), c_i(Nz), c_i(Nr))


# ------------------------------------------------------------------ spectral
def filter_(F, fz, fr):
    Nz, Nr = F.shape
    lib().orc_filter(_c128(F), _f64(fz), _f64(fr), c_i(Nz), c_i(Nr))


def rt_to_pm(r, t, p, m):
    lib().orc_rt_to_pm(_c128(r), _c128(t), _c128(p), _c128(m), c_l(r.size))


def pm_to_rt(p, m, r, t):
    lib().orc_pm_to_rt(_c128(p), _c128(m), _c128(r), _c128(t), c_l(r.size))


def correct_currents_curlfree(rho_prev, rho_next, Jp, Jm, Jz, kz, kr, inv_k2, inv_dt):
    Nz, Nr = Jp.shape
    lib().orc_correct_currents_curlfree(_c128(rho_prev), _c128(rho_next), _c128(Jp), _c128(Jm),
                                        _c128(Jz), _f64(kz), _f64(kr), _f64(inv_k2),
                                        c_d(inv_dt), c_i(Nz), c_i(Nr))


def push_eb_standard(Ep, Em, Ez, Bp, Bm, Bz, Jp, Jm, Jz, rho_prev, rho_next,
                     rho_prev_coef, rho_next_coef, j_coef, C, S_w, kr, kz, dt, use_true_rho):
    Nz, Nr = Ep.shape
    lib().orc_push_eb_standard(_c128(Ep), _c128(Em), _c128(Ez), _c128(Bp), _c128(Bm), _c128(Bz),
                               _c128(Jp), _c128(Jm), _c128(Jz), _c128(rho_prev), _c128(rho_next),
                               _f64(rho_prev_coef), _f64(rho_next_coef), _f64(j_coef),
                               _f64(C), _f64(S_w), _f64(kr), _f64(kz), c_d(dt),
                               c_i(int(use_true_rho)), c_i(Nz), c_i(Nr))


def correct_currents_curlfree_comoving(rho_prev, rho_next, Jp, Jm, Jz, kz, kr, inv_k2,
                                       j_corr_coef, T_eb, T_cc):
    Nz, Nr = Jp.shape
    lib().orc_correct_currents_curlfree_comoving(
        _c128(rho_prev), _c128(rho_next), _c128(Jp), _c128(Jm), _c128(Jz), _f64(kz), _f64(kr),
        _f64(inv_k2), _c128(j_corr_coef), _c128(T_eb), _c128(T_cc), c_i(Nz), c_i(Nr))


def correct_currents_crossdeposition(rho_prev, rho_next, rho_next_z, rho_next_xy, Jp, Jm, Jz,
                                     kz, kr, inv_dt):
    Nz, Nr = Jp.shape
    lib().orc_correct_currents_crossdeposition(
        _c128(rho_prev), _c128(rho_next), _c128(rho_next_z), _c128(rho_next_xy), _c128(Jp),
        _c128(Jm), _c128(Jz), _f64(kz), _f64(kr), c_d(inv_dt), c_i(Nz), c_i(Nr))


def correct_currents_crossdeposition_comoving(rho_prev, rho_next, rho_next_z, rho_next_xy,
                                              Jp, Jm, Jz, kz, kr, j_corr_coef, T_eb, T_cc):
    Nz, Nr = Jp.shape
    lib().orc_correct_currents_crossdeposition_comoving(
        _c128(rho_prev), _c128(rho_next), _c128(rho_next_z), _c128(rho_next_xy), _c128(Jp),
        _c128(Jm), _c128(Jz), _f64(kz), _f64(kr), _c128(j_corr_coef), _c128(T_eb), _c128(T_cc),
        c_i(Nz), c_i(Nr))


def push_eb_comoving(Ep, Em, Ez, Bp, Bm, Bz, Jp, Jm, Jz, rho_prev, rho_next,
                     rho_prev_coef, rho_next_coef, j_coef, C, S_w, T_eb, T_cc, T_rho, kr, kz,
                     dt, V, use_true_rho):
    Nz, Nr = Ep.shape
    lib().orc_push_eb_comoving(
        _c128(Ep), _c128(Em), _c128(Ez), _c128(Bp), _c128(Bm), _c128(Bz), _c128(Jp), _c128(Jm),
        _c128(Jz), _c128(rho_prev), _c128(rho_next), _c128(rho_prev_coef), _c128(rho_next_coef),
        _c128(j_coef), _f64(C), _f64(S_w), _c128(T_eb), _c128(T_cc), _c128(T_rho), _f64(kr),
        _f64(kz), c_d(dt), c_d(V), c_i(int(use_true_rho)), c_i(Nz), c_i(Nr))


# Threads of the z-FFT.  The reference's CPU path plans its FFTW transforms with `threads = nthreads`
# (fourier.py:59-96); 1 = np.fft as before.  bench.py's cpu_baseline sets it to the core count: the
# columns of a (Nz, Nr) grid are then transformed by a thread pool (scipy.fft `workers`, the same
# pocketfft kernels - the results are bit-identical to np.fft, checked in tests/test_oracle_golden.py).
FFT_WORKERS = 1


def fft_z(a):
    """fourier.py:104-126: unnormalised forward DFT along axis 0."""
    if FFT_WORKERS > 1:
        import scipy.fft
        return scipy.fft.fft(a, axis=0, workers=FFT_WORKERS)
    return np.fft.fft(a, axis=0)


def ifft_z(a):
    """fourier.py:128-168: backward DFT along axis 0 including the 1/Nz factor."""
    if FFT_WORKERS > 1:
        import scipy.fft
        return scipy.fft.ifft(a, axis=0, workers=FFT_WORKERS)
    return np.fft.ifft(a, axis=0)


def dht(F, mat):
    """hankel.py:182-243: complex (Nz,Nr) -> real (2Nz,Nr) [re rows, then im rows],
    one real matrix product with the (Nr,Nr) matrix, back to complex."""
    Nz = F.shape[0]
    a = np.empty((2 * Nz, F.shape[1]))
    a[:Nz] = F.real
    a[Nz:] = F.imag
    o = np.dot(a, mat)
    return o[:Nz] + 1.j * o[Nz:]


class Transformer:
    """spectral_transform/spectral_transformer.py:89-223 for one azimuthal mode.
    mats = dict(M0, invM0, Mp, invMp, Mm, invMm) from the host-side DHT setup."""

    def __init__(self, mats):
        self.m = mats

    def interp2spect_scal(self, a):
        return dht(fft_z(a), self.m['M0'])

    def spect2interp_scal(self, a):
        return ifft_z(dht(a, self.m['invM0']))

    def interp2spect_vect(self, r, t):
        br = fft_z(r)
        bt = fft_z(t)
        p = np.empty_like(br)
        mm = np.empty_like(br)
        rt_to_pm(np.ascontiguousarray(br), np.ascontiguousarray(bt), p, mm)
        return dht(p, self.m['Mp']), dht(mm, self.m['Mm'])

    def spect2interp_vect(self, p, mm):
        bp = np.ascontiguousarray(dht(p, self.m['invMp']))
        bm = np.ascontiguousarray(dht(mm, self.m['invMm']))
        r = np.empty_like(bp)
        t = np.empty_like(bp)
        pm_to_rt(bp, bm, r, t)
        return ifft_z(r), ifft_z(t)


INTERP = ['Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz', 'Jr', 'Jt', 'Jz', 'rho']
SPECT = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']


class OracleSim:
    """Whole PIC cycle on the CPU, restating Simulation.step / .deposit /
    .exchange_and_damp_EB (fbpic/main.py:346-769) for the single-domain, z-periodic,
    standard-PSATD, curl-free configuration (C1/C2/C5).

    tables: host-side setup arrays (the product's fbpic_amd host code builds them and
    they are themselves pinned against tests/golden/grid_setup.npz):
      per mode m: M0,invM0,Mp,invMp,Mm,invMm, kz(Nz,Nr), kr(Nz,Nr), inv_k2, filter_z,
      filter_r, C, S_w, j_coef, rho_prev_coef, rho_next_coef, invvol, ruyten (Nr+1)
    species: list of dicts with q, m, and the 8 particle arrays.
    """

    def __init__(self, Nz, Nr, Nm, zmin, zmax, rmax, dt, shape, tables, species,
                 nthreads=1, filter_currents=True, v_comoving=None, use_galilean=False,
                 current_correction='curl-free'):
        # Galilean / comoving-current PSATD (main.py:269-273, 496-497, 524-525): tables then
        # also hold T_eb, T_cc, T_rho, j_corr_coef and complex source coefficients
        self.v_comoving = v_comoving
        self.use_galilean = bool(use_galilean) if v_comoving is not None else False
        self.current_correction = current_correction
        self.Nz, self.Nr, self.Nm = Nz, Nr, Nm
        self.zmin, self.zmax, self.rmax = zmin, zmax, rmax
        self.dz = (zmax - zmin) / Nz
        self.dr = rmax / Nr
        self.invdz = 1. / self.dz
        self.invdr = 1. / self.dr
        self.dt = dt
        self.shape = shape
        self.t = tables
        self.nthreads = nthreads
        self.filter_currents = filter_currents
        self.trans = [Transformer(tables[m]) for m in range(Nm)]
        z = lambda: np.zeros((Nz, Nr), dtype=np.complex128)  # noqa: E731
        self.interp = [{k: z() for k in INTERP} for _ in range(Nm)]
        self.spect = [{k: z() for k in SPECT} for _ in range(Nm)]
        if current_correction == 'cross-deposition':      # spectral_grid.py:97-99
            for sp in self.spect:
                sp['rho_next_z'] = z()
                sp['rho_next_xy'] = z()
        self.species = species
        for s in species:
            n = s['x'].size
            for k in ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz'):
                s.setdefault(k, np.zeros(n))
        self.time = 0.
        self.iteration = 0
        self.glob = [np.zeros((nthreads, Nm, Nz + 4, Nr + 4), dtype=np.complex128)
                     for _ in range(3)]
        # wall seconds per phase of step() (bench.py's cpu_baseline reports them)
        self.phase_seconds = {}

    def _ph(self, name):
        sim = self

        class _T:
            def __enter__(self):
                self.t = time.perf_counter()

            def __exit__(self, *exc):
                sim.phase_seconds[name] = sim.phase_seconds.get(name, 0.) + time.perf_counter() - self.t
        return _T()

    # -- transforms (fields/fields.py:313-429)
    def interp2spect(self, ft):
        for m in range(self.Nm):
            it, sp, tr = self.interp[m], self.spect[m], self.trans[m]
            if ft in ('E', 'B', 'J'):
                sp[ft + 'z'][:] = tr.interp2spect_scal(it[ft + 'z'])
                sp[ft + 'p'][:], sp[ft + 'm'][:] = tr.interp2spect_vect(it[ft + 'r'], it[ft + 't'])
            else:
                sp[ft][:] = tr.interp2spect_scal(it['rho'])

    def spect2interp(self, ft):
        for m in range(self.Nm):
            it, sp, tr = self.interp[m], self.spect[m], self.trans[m]
            if ft in ('E', 'B', 'J'):
                it[ft + 'z'][:] = tr.spect2interp_scal(sp[ft + 'z'])
                it[ft + 'r'][:], it[ft + 't'][:] = tr.spect2interp_vect(sp[ft + 'p'], sp[ft + 'm'])
            else:
                it['rho'][:] = tr.spect2interp_scal(sp[ft])

    def partial_roundtrip(self, ft):
        """spect2partial_interp followed by partial_interp2spect
        (fields/fields.py:431-536; main.py:741-766): iFFT then FFT along z."""
        for m in range(self.Nm):
            sp = self.spect[m]
            for k in (ft + 'z', ft + 'p', ft + 'm'):
                sp[k][:] = fft_z(ifft_z(sp[k]))

    # -- deposition (main.py:588-670)
    def deposit(self, fieldtype):
        t = self.t
        Nm, Nz, Nr = self.Nm, self.Nz, self.Nr
        ruy = 'ruyten_linear' if self.shape == 'linear' else 'ruyten_cubic'
        b0 = t[0][ruy]
        bh = t[1 if Nm > 1 else 0][ruy]
        geom = (self.invdz, self.zmin, Nz, self.invdr, 0., Nr)
        if fieldtype.startswith('rho'):
            with self._ph('erase'):
                for m in range(Nm):
                    self.interp[m]['rho'][:] = 0.
                self.glob[0][:] = 0.
            with self._ph('deposit'):
                for s in self.species:
                    if s['q'] == 0:
                        continue
                    deposit_rho_global(self.shape, Nm, s['x'], s['y'], s['z'], s['w'], s['q'],
                                       *geom, b0, bh, self.nthreads, self.glob[0])
            with self._ph('reduce'):
                for m in range(Nm):
                    sum_reduce(self.glob[0], m, self.interp[m]['rho'])
                    divide_by_volume(self.interp[m]['rho'], t[m]['invvol'])
        else:
            with self._ph('erase'):
                for g in self.glob:
                    g[:] = 0.
                for m in range(Nm):
                    for k in ('Jr', 'Jt', 'Jz'):
                        self.interp[m][k][:] = 0.
            with self._ph('deposit'):
                for s in self.species:
                    if s['q'] == 0:
                        continue
                    deposit_J_global(self.shape, Nm, s['x'], s['y'], s['z'], s['w'], s['q'],
                                     s['ux'], s['uy'], s['uz'], s['inv_gamma'], *geom, b0, bh,
                                     self.nthreads, self.glob)
            with self._ph('reduce'):
                for m in range(Nm):
                    for i, k in enumerate(('Jr', 'Jt', 'Jz')):
                        sum_reduce(self.glob[i], m, self.interp[m][k])
                        divide_by_volume(self.interp[m][k], t[m]['invvol'])
        with self._ph('transforms'):
            self.interp2spect(fieldtype)
        if self.filter_currents:
            with self._ph('field kernels'):
                for m in range(Nm):
                    sp = self.spect[m]
                    keys = ('Jp', 'Jm', 'Jz') if fieldtype == 'J' else (fieldtype,)
                    for k in keys:
                        filter_(sp[k], t[m]['filter_z'], t[m]['filter_r'])

    def gather(self):
        grids = [[self.interp[m][k] for k in INTERP[:6]] for m in range(self.Nm)]
        for s in self.species:
            if s['q'] == 0:
                continue
            gather(self.shape, self.Nm == 2, s['x'], s['y'], s['z'], self.rmax,
                   self.invdz, self.zmin, self.Nz, self.invdr, 0., self.Nr, grids,
                   s['Ex'], s['Ey'], s['Ez'], s['Bx'], s['By'], s['Bz'])

    def shift_galilean_boundaries(self, dt):
        """main.py:772-790: only the grid position moves."""
        self.zmin += self.v_comoving * dt
        self.zmax += self.v_comoving * dt

    def cross_deposit(self):
        """main.py:672-716, called with the particles at t = n+1/2: deposit the charge at
        (z[n], x[n+1]) and at (z[n+1], x[n]), then come back to n+1/2."""
        dt = self.dt

        def push(frac, px, py, pz):
            for s in self.species:
                push_x(s['x'], s['y'], s['z'], s['ux'], s['uy'], s['uz'], s['inv_gamma'],
                       frac * dt, px, py, pz)
        push(0.5, 1., 1., -1.)
        if self.use_galilean:
            self.shift_galilean_boundaries(-0.5 * dt)
        self.deposit('rho_next_xy')
        push(1., -1., -1., 1.)
        if self.use_galilean:
            self.shift_galilean_boundaries(dt)
        self.deposit('rho_next_z')
        push(0.5, 1., 1., -1.)
        if self.use_galilean:
            self.shift_galilean_boundaries(-0.5 * dt)

    def step(self, N=1, correct_currents=True, use_true_rho=False):
        dt = self.dt
        self.interp2spect('E')
        self.interp2spect('B')
        for i_step in range(N):
            # exchange_period == 1 for single-proc periodic (boundary_communicator.py:294)
            for s in self.species:
                shift_periodic(s['z'], self.zmin, self.zmax)
            self.deposit('rho_prev')
            if i_step == 0:
                self.deposit('J')
            with self._ph('gather'):
                self.gather()
            with self._ph('push'):
                for s in self.species:
                    if s['q'] != 0:
                        push_p(s['ux'], s['uy'], s['uz'], s['inv_gamma'], s['Ex'], s['Ey'], s['Ez'],
                               s['Bx'], s['By'], s['Bz'], s['q'], s['m'], dt)
                for s in self.species:
                    push_x(s['x'], s['y'], s['z'], s['ux'], s['uy'], s['uz'], s['inv_gamma'], 0.5 * dt)
            if self.use_galilean:
                self.shift_galilean_boundaries(0.5 * dt)
            self.deposit('J')
            cross = correct_currents and self.current_correction == 'cross-deposition'
            if cross:
                self.cross_deposit()
            with self._ph('push'):
                for s in self.species:
                    push_x(s['x'], s['y'], s['z'], s['ux'], s['uy'], s['uz'], s['inv_gamma'], 0.5 * dt)
            if self.use_galilean:
                self.shift_galilean_boundaries(0.5 * dt)
            self.deposit('rho_next')
            t_field = time.perf_counter()
            V = self.v_comoving
            if cross:
                for m in range(self.Nm):
                    sp, t = self.spect[m], self.t[m]
                    if V is None:
                        correct_currents_crossdeposition(
                            sp['rho_prev'], sp['rho_next'], sp['rho_next_z'], sp['rho_next_xy'],
                            sp['Jp'], sp['Jm'], sp['Jz'], t['kz'], t['kr'], 1. / dt)
                    else:
                        correct_currents_crossdeposition_comoving(
                            sp['rho_prev'], sp['rho_next'], sp['rho_next_z'], sp['rho_next_xy'],
                            sp['Jp'], sp['Jm'], sp['Jz'], t['kz'], t['kr'], t['j_corr_coef'],
                            t['T_eb'], t['T_cc'])
            elif correct_currents:
                for m in range(self.Nm):
                    sp, t = self.spect[m], self.t[m]
                    if V is None:
                        correct_currents_curlfree(sp['rho_prev'], sp['rho_next'], sp['Jp'], sp['Jm'],
                                                  sp['Jz'], t['kz'], t['kr'], t['inv_k2'], 1. / dt)
                    else:
                        correct_currents_curlfree_comoving(
                            sp['rho_prev'], sp['rho_next'], sp['Jp'], sp['Jm'], sp['Jz'], t['kz'],
                            t['kr'], t['inv_k2'], t['j_corr_coef'], t['T_eb'], t['T_cc'])
            for m in range(self.Nm):
                sp, t = self.spect[m], self.t[m]
                if V is None:
                    push_eb_standard(sp['Ep'], sp['Em'], sp['Ez'], sp['Bp'], sp['Bm'], sp['Bz'],
                                     sp['Jp'], sp['Jm'], sp['Jz'], sp['rho_prev'], sp['rho_next'],
                                     t['rho_prev_coef'], t['rho_next_coef'], t['j_coef'],
                                     t['C'], t['S_w'], t['kr'], t['kz'], dt, use_true_rho)
                else:
                    push_eb_comoving(sp['Ep'], sp['Em'], sp['Ez'], sp['Bp'], sp['Bm'], sp['Bz'],
                                     sp['Jp'], sp['Jm'], sp['Jz'], sp['rho_prev'], sp['rho_next'],
                                     t['rho_prev_coef'], t['rho_next_coef'], t['j_coef'], t['C'],
                                     t['S_w'], t['T_eb'], t['T_cc'], t['T_rho'], t['kr'], t['kz'],
                                     dt, V, use_true_rho)
                sp['rho_prev'][:] = sp['rho_next']
                sp['rho_next'][:] = 0.
            self.phase_seconds['field kernels'] = self.phase_seconds.get('field kernels', 0.) \
                + time.perf_counter() - t_field
            with self._ph('transforms'):
                self.partial_roundtrip('E')
                self.partial_roundtrip('B')
                self.spect2interp('E')
                self.spect2interp('B')
            self.time += dt
            self.iteration += 1
        self.spect2interp('J')
        self.spect2interp('rho_prev')


PTCL = ['x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w', 'Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz']


def tables_from_sim(sim):
    """Host setup tables (NumPy) of an fbpic_amd `Simulation` in the form OracleSim
    expects.  Only host-side setup data is read (matrices, coefficient tables, volumes);
    those tables are pinned against the reference in tests/test_host_setup.py."""
    fld = sim.fld
    tabs = []
    for m in range(fld.Nm):
        tr, sp, ps, it = fld.trans[m], fld.spect[m], fld.psatd[m], fld.interp[m]
        tabs.append(dict(
            M0=tr.dht0.M, invM0=tr.dht0.invM, Mp=tr.dhtp.M, invMp=tr.dhtp.invM,
            Mm=tr.dhtm.M, invMm=tr.dhtm.invM,
            kz=np.ascontiguousarray(sp.kz), kr=np.ascontiguousarray(sp.kr),
            inv_k2=np.ascontiguousarray(sp.inv_k2),
            filter_z=sp.filter_array_z, filter_r=sp.filter_array_r,
            C=ps.C, S_w=ps.S_w, j_coef=ps.j_coef, rho_prev_coef=ps.rho_prev_coef,
            rho_next_coef=ps.rho_next_coef, invvol=it.invvol,
            ruyten_linear=it.ruyten_linear_coef, ruyten_cubic=it.ruyten_cubic_coef))
        if ps.V is not None:
            for k in ('j_coef', 'rho_prev_coef', 'rho_next_coef', 'T_eb', 'T_cc', 'T_rho',
                      'j_corr_coef'):
                tabs[-1][k] = np.ascontiguousarray(getattr(ps, k), dtype=np.complex128)
    return tabs


def from_sim(sim, nthreads=1):
    """OracleSim holding a private host copy of the state of `sim` (which must be on the
    host): same inputs, independent compute path."""
    fld = sim.fld
    g0 = fld.interp[0]
    species = []
    for s in sim.ptcl:
        d = dict(q=s.q, m=s.m)
        for k in PTCL:
            d[k] = np.array(getattr(s, k), dtype=np.float64, copy=True)
        species.append(d)
    o = OracleSim(fld.Nz, fld.Nr, fld.Nm, g0.zmin, g0.zmax, fld.rmax, sim.dt,
                  sim.particle_shape, tables_from_sim(sim), species, nthreads=nthreads,
                  filter_currents=sim.filter_currents, v_comoving=sim.v_comoving,
                  use_galilean=sim.use_galilean,
                  current_correction=fld.current_correction)
    for m in range(fld.Nm):
        for k in INTERP:
            o.interp[m][k][:] = getattr(fld.interp[m], k)
    return o
