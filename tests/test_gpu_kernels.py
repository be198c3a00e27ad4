"""Kernel-level parity: every C-ABI entry point of libfbpic_amd.so on the MI355X against
the CPU oracle (oracle/, itself pinned to the reference) and the golden vectors.

Tolerances: cell indices bit-exact; push bit-exact against the oracle (same
operation order, no FMA contraction, IEEE sqrt/div); gather / field kernels 1e-13 of
max|array|; deposition / transforms 1e-13 * max|F| (the reference's own CPU<->GPU bound,
tests/test_cpu_gpu_deposition.py:96).
"""
import ctypes
import numpy as np
import pytest
from scipy.constants import c, e, m_e, epsilon_0, mu_0
from conftest import golden, rel_err, achieved

pytestmark = pytest.mark.gpu
TOL = 1e-13
TOL_GATHER = 1e-14
TOL_TRANSFORM = 1e-13


@pytest.fixture(scope='module')
def hip():
    from fbpic_amd import _capi
    _capi.require_device()
    return _capi


def dev(hip, a, dtype=None):
    return hip.to_device(np.ascontiguousarray(a, dtype=dtype))


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ push
@pytest.mark.parametrize('n', [4096, 4095, 1])
def test_push_bit_exact(hip, oracle, n):
    g = golden('push')
    dt = float(g['dt'])
    names = ('ux', 'uy', 'uz', 'inv_gamma', 'Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz', 'x', 'y', 'z')
    h = {k: g['in_' + k][:n].copy() for k in names}
    d = {k: dev(hip, h[k]) for k in names}
    p = hip.ptr
    rc = hip.lib().fb_push_p(n, p(d['ux']), p(d['uy']), p(d['uz']), p(d['inv_gamma']),
                             p(d['Ex']), p(d['Ey']), p(d['Ez']), p(d['Bx']), p(d['By']), p(d['Bz']),
                             -e, m_e, c, dt, hip.stream())
    hip.check(rc, 'fb_push_p')
    oracle.push_p(h['ux'], h['uy'], h['uz'], h['inv_gamma'], h['Ex'], h['Ey'], h['Ez'],
                  h['Bx'], h['By'], h['Bz'], -e, m_e, dt)
    for k in ('ux', 'uy', 'uz', 'inv_gamma'):
        assert np.array_equal(host(d[k]), h[k]), k
        ref = g['pp_e_' + k][:n]
        assert np.all(np.abs(h[k] - ref) <= 4e-15 * np.abs(ref) + 1e-300)
    rc = hip.lib().fb_push_x(n, p(d['x']), p(d['y']), p(d['z']), p(d['ux']), p(d['uy']), p(d['uz']),
                             p(d['inv_gamma']), c, 0.5 * dt, 1., 1., 1., hip.stream())
    hip.check(rc, 'fb_push_x')
    for k, gk in (('x', 'px_x'), ('y', 'px_y'), ('z', 'px_z')):
        assert np.array_equal(host(d[k]), g[gk][:n]), k   # bit-exact vs the reference itself


def test_array_wrapper_family(hip, oracle):
    """fb_malloc / fb_h2d / fb_d2h / fb_free / fb_last_error_string (the reference's
    `cuda.to_device` / `copy_to_host`, utils/cuda.py:101-137): a position push on arrays that never were
    PyTorch tensors, bit-exact against the oracle on the golden inputs (tests/golden/push.npz)."""
    import ctypes
    lib = hip.lib()
    g = golden('push')
    n = 4096
    names = ('x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma')
    h = {k: np.ascontiguousarray(g['in_' + k][:n]) for k in names}
    d = {}
    for k in names:
        ptr = ctypes.c_void_p()
        hip.check(lib.fb_malloc(8 * n, ctypes.byref(ptr)), 'fb_malloc')
        assert ptr.value
        hip.check(lib.fb_h2d(ptr, h[k].ctypes.data_as(ctypes.c_void_p), 8 * n, hip.stream()), 'fb_h2d')
        d[k] = ptr
    hip.check(lib.fb_push_x(n, d['x'], d['y'], d['z'], d['ux'], d['uy'], d['uz'], d['inv_gamma'], c,
                            0.5 * float(g['dt']), 1., 1., 1., hip.stream()), 'fb_push_x')
    out = {k: np.empty(n) for k in ('x', 'y', 'z')}
    for k in out:
        hip.check(lib.fb_d2h(out[k].ctypes.data_as(ctypes.c_void_p), d[k], 8 * n, hip.stream()), 'fb_d2h')
    hip.check(lib.fb_sync(hip.stream()), 'fb_sync')
    ref = {k: h[k].copy() for k in ('x', 'y', 'z')}
    oracle.push_x(ref['x'], ref['y'], ref['z'], h['ux'], h['uy'], h['uz'], h['inv_gamma'], 0.5 * float(g['dt']), 1., 1., 1.)
    for k in ('x', 'y', 'z'):
        assert np.array_equal(out[k], ref[k]), k
    for k in names:
        hip.check(lib.fb_free(d[k]), 'fb_free')
    # zero bytes and NULL are accepted; a failing call leaves its text in both error getters
    z = ctypes.c_void_p(1)
    assert lib.fb_malloc(0, ctypes.byref(z)) == 0 and not z.value and lib.fb_free(None) == 0
    assert lib.fb_malloc(8, None) != 0
    assert lib.fb_last_error_string() == lib.fb_last_error() and b'fb_malloc' in lib.fb_last_error_string()


def test_shift_periodic(hip, oracle):
    rng = np.random.default_rng(5)
    z = rng.uniform(-3., 4., 10001)
    dz = dev(hip, z)
    hip.check(hip.lib().fb_shift_periodic(z.size, hip.ptr(dz), 0.25, 1.5, hip.stream()), 'shift')
    oracle.shift_periodic(z, 0.25, 1.5)
    assert np.array_equal(host(dz), z)
    assert z.min() >= 0.25 and z.max() < 1.5


# ------------------------------------------------------------------ gather
def _gather_gpu(hip, g, shape, nm, slab):
    Nz, Nr = int(g['Nz']), int(g['Nr'])
    n = g['x'].size
    t = hip.torch()
    if slab:   # z-major slab views (row stride 6*nm*Nr), as used by Fields
        s = t.empty((Nz, 6 * nm, Nr), dtype=t.complex128, device='cuda')
        for m in range(nm):
            for k in range(6):
                s[:, 6 * m + k, :] = dev(hip, g['grids'][m, k])
        views = [s[:, j, :] for j in range(6 * nm)]
    else:
        views = [dev(hip, g['grids'][m, k]) for m in range(nm) for k in range(6)]
    x, y, z = dev(hip, g['x']), dev(hip, g['y']), dev(hip, g['z'])
    F = [t.full((n,), 7., dtype=t.float64, device='cuda') for _ in range(6)]
    p = hip.ptr
    rc = hip.lib().fb_gather(1 if shape == 'linear' else 3, nm, n, p(x), p(y), p(z),
                             float(g['rmax_gather']), 1. / float(g['dz']), float(g['zmin']), Nz,
                             1. / float(g['dr']), 0., Nr, hip.ptr_array(views),
                             hip.row_stride(views[0]), *[p(f) for f in F], hip.stream())
    hip.check(rc, 'fb_gather')
    return np.array([host(f) for f in F])


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
@pytest.mark.parametrize('slab', [False, True])
def test_gather(hip, oracle, shape, slab):
    g = golden('gather')
    got = _gather_gpu(hip, g, shape, 2, slab)
    achieved('gather %s Nm=2 vs reference' % shape, rel_err(got, g['%s_nm2' % shape]), TOL_GATHER)
    for nm in (1, 3, 4):
        got = _gather_gpu(hip, g, shape, nm, slab)
        achieved('gather %s Nm=%d vs reference' % (shape, nm),
                 rel_err(got, g['%s_nm%d_onemode' % (shape, nm)]), TOL_GATHER)
    out = np.hypot(g['x'], g['y']) >= float(g['rmax_gather'])
    assert np.all(got[:, out] == 0.)


# ------------------------------------------------------------------ cell index / sort
def _sort(hip, x, y, z, geom, Nz, Nr):
    t = hip.torch()
    n = x.size
    ncell = Nz * (Nr + 1)
    dx, dy, dz_ = dev(hip, x), dev(hip, y), dev(hip, z)
    ci = t.empty(n, dtype=t.int32, device='cuda')
    si = t.empty(n, dtype=t.int32, device='cuda')
    pre = t.full((ncell,), -1, dtype=t.int32, device='cuda')
    p = hip.ptr
    hip.check(hip.lib().fb_cell_index(n, p(dx), p(dy), p(dz_), *geom, p(ci), p(si), hip.stream()), 'ci')
    ci0 = host(ci).copy()
    nb = int(hip.lib().fb_sort_workspace_bytes(n, ncell))
    ws = t.empty(nb, dtype=t.uint8, device='cuda')
    ci2, si2 = t.empty_like(ci), t.empty_like(si)
    in_alt = ctypes.c_int(-1)
    hip.check(hip.lib().fb_sort_by_cell(n, ncell, p(ci), p(si), p(ci2), p(si2), ctypes.byref(in_alt),
                                        p(pre), p(ws), nb, hip.stream()), 'sort')
    assert in_alt.value in (0, 1)
    if in_alt.value:
        ci, si = ci2, si2
    return ci0, host(ci), host(si), host(pre), (dx, dy, dz_, si)


def test_cell_index_and_sort(hip, oracle):
    g = golden('deposit')
    Nz, Nr = int(g['Nz']), int(g['Nr'])
    geom = (1. / float(g['dz']), float(g['zmin']), Nz, 1. / float(g['dr']), 0., Nr)
    ci0, cis, si, pre, (dx, dy, dz_, dsi) = _sort(hip, g['x'], g['y'], g['z'], geom, Nz, Nr)
    ref = oracle.cell_index(g['x'], g['y'], g['z'], *geom)
    assert np.array_equal(ci0, ref)                      # bit-exact particle -> cell index
    order = np.argsort(ref, kind='stable')               # Thrust argsort is stable too
    assert np.array_equal(si, order.astype(np.int32))
    assert np.array_equal(cis, ref[order])
    counts = np.bincount(ref, minlength=Nz * (Nr + 1))
    assert np.array_equal(pre, np.cumsum(counts).astype(np.int32))
    # permutation of several attributes in one launch
    t = hip.torch()
    src = [dx, dy, dz_]
    dst = [t.empty_like(dx) for _ in range(3)]
    hip.check(hip.lib().fb_permute(dx.shape[0], hip.ptr(dsi), 3, hip.ptr_array(src),
                                   hip.ptr_array(dst), hip.stream()), 'permute')
    for a, b in zip((g['x'], g['y'], g['z']), dst):
        assert np.array_equal(host(b), a[order])


def test_cell_index_large_random(hip, oracle):
    rng = np.random.default_rng(11)
    n, Nz, Nr = 1 << 20, 256, 64
    dzc = 0.2e-6
    r = rng.uniform(0, 1.05 * Nr * dzc, n)
    th = rng.uniform(0, 2 * np.pi, n)
    x, y = r * np.cos(th), r * np.sin(th)
    z = rng.uniform(0, Nz * dzc, n)
    geom = (1. / dzc, 0., Nz, 1. / dzc, 0., Nr)
    ci0, cis, si, pre, _ = _sort(hip, x, y, z, geom, Nz, Nr)
    ref = oracle.cell_index(x, y, z, *geom)
    assert np.array_equal(ci0, ref)
    assert np.all(np.diff(cis) >= 0) and pre[-1] == n
    assert np.array_equal(np.sort(si), np.arange(n, dtype=np.int32))


# ------------------------------------------------------------------ deposition
def _deposit_gpu(hip, g, shape, Nm, what, b0, bh, slab=False, presort=True):
    Nz, Nr = int(g['Nz']), int(g['Nr'])
    geom = (1. / float(g['dz']), float(g['zmin']), Nz, 1. / float(g['dr']), 0., Nr)
    t = hip.torch()
    _, _, si, pre, _ = _sort(hip, g['x'], g['y'], g['z'], geom, Nz, Nr)
    names = ('x', 'y', 'z', 'w', 'ux', 'uy', 'uz', 'inv_gamma')
    if not presort:                                       # result must not depend on the order
        si = np.random.default_rng(1).permutation(g['x'].size)
    d = {k: dev(hip, g[k][si]) for k in names}            # (sorted) particle arrays
    dpre = dev(hip, pre)
    ncomp = 1 if what == 'rho' else 3
    if slab == 'records':      # node-major target: all fields of a node in one record
        rec = ncomp * Nm + 1
        s = t.zeros((Nz, Nr, rec), dtype=t.complex128, device='cuda')
        views = [s[:, :, j] for j in range(ncomp * Nm)]
    elif slab:
        s = t.zeros((Nz, ncomp * Nm + 1, Nr), dtype=t.complex128, device='cuda')
        views = [s[:, j, :] for j in range(ncomp * Nm)]
    else:
        views = [t.zeros((Nz, Nr), dtype=t.complex128, device='cuda') for _ in range(ncomp * Nm)]
    p = hip.ptr
    sh = 1 if shape == 'linear' else 3
    db0, dbh = dev(hip, b0), dev(hip, bh)
    if what == 'rho':
        rc = hip.lib().fb_deposit_rho(sh, Nm, g['x'].size, p(d['x']), p(d['y']), p(d['z']), p(d['w']),
                                      float(g['q']), *geom, hip.ptr_array(views),
                                      views[0].stride(0), views[0].stride(1), p(dpre), p(db0), p(dbh),
                                      None, hip.stream())
    else:
        rc = hip.lib().fb_deposit_J(sh, Nm, g['x'].size, p(d['x']), p(d['y']), p(d['z']), p(d['w']),
                                    float(g['q']), p(d['ux']), p(d['uy']), p(d['uz']),
                                    p(d['inv_gamma']), c, *geom, hip.ptr_array(views),
                                    views[0].stride(0), views[0].stride(1), p(dpre), p(db0), p(dbh),
                                    None, hip.stream())
    hip.check(rc, 'fb_deposit')
    return np.array([host(v) for v in views])


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
@pytest.mark.parametrize('Nm', [1, 2, 3, 4, 5])
def test_deposit_vs_oracle_and_golden(hip, oracle, shape, Nm):
    g = golden('deposit')
    Nz, Nr = int(g['Nz']), int(g['Nr'])
    geom = (1. / float(g['dz']), float(g['zmin']), Nz, 1. / float(g['dr']), 0., Nr)
    sh = 'lin' if shape == 'linear' else 'cub'
    for ruy in (1, 0):
        b0 = g['ruy_%s_m0' % sh] if ruy else np.zeros(Nr + 1)
        bh = (g['ruy_%s_m1' % sh] if Nm > 1 else b0) if ruy else np.zeros(Nr + 1)
        # oracle
        glob = oracle.deposit_rho_global(shape, Nm, g['x'], g['y'], g['z'], g['w'], float(g['q']),
                                         *geom, b0, bh, 1)
        red = np.zeros((Nm, Nz, Nr), complex)
        for m in range(Nm):
            oracle.sum_reduce(glob, m, red[m])
        got = _deposit_gpu(hip, g, shape, Nm, 'rho', b0, bh, slab=bool(ruy))
        assert rel_err(got, red) < TOL
        got_u = _deposit_gpu(hip, g, shape, Nm, 'rho', b0, bh, slab=bool(ruy), presort=False)
        assert rel_err(got_u, red) < TOL
        got_r = _deposit_gpu(hip, g, shape, Nm, 'rho', b0, bh, slab='records')
        assert rel_err(got_r, red) < TOL
        gl = oracle.deposit_J_global(shape, Nm, g['x'], g['y'], g['z'], g['w'], float(g['q']),
                                     g['ux'], g['uy'], g['uz'], g['inv_gamma'], *geom, b0, bh, 1)
        gotJr = _deposit_gpu(hip, g, shape, Nm, 'J', b0, bh, slab='records').reshape(Nm, 3, Nz, Nr)
        gotJ = _deposit_gpu(hip, g, shape, Nm, 'J', b0, bh, slab=bool(ruy)).reshape(Nm, 3, Nz, Nr)
        assert rel_err(gotJr, gotJ) < TOL
        for k in range(3):
            redk = np.zeros((Nm, Nz, Nr), complex)
            for m in range(Nm):
                oracle.sum_reduce(gl[k], m, redk[m])
            assert rel_err(gotJ[:, k], redk) < TOL, (k, ruy)
        if Nm in (1, 2, 4):   # and directly against the reference's output
            tag = '%s_r%d_nm%d' % (shape, ruy, Nm)
            assert rel_err(got, g['rho_' + tag]) < TOL
            for k in range(3):
                assert rel_err(gotJ[:, k], g['J_' + tag][k]) < TOL


# ------------------------------------------------------------------ grid kernels
def test_spectral_kernels(hip, oracle):
    g = golden('spectral')
    gs = golden('grid_setup')
    Nm, Nz, Nr = int(g['Nm']), int(g['Nz']), int(g['Nr'])
    dt = float(g['dt'])
    names = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']
    t = hip.torch()
    p = hip.ptr
    for m in range(Nm):
        tg = 'o-1_m%d' % m
        kz = np.repeat(gs['kz_' + tg][:, None], Nr, 1).copy()
        kr = np.repeat(gs['kr_' + tg][None, :], Nz, 0).copy()
        dkz, dkr, dik2 = dev(hip, kz), dev(hip, kr), dev(hip, gs['inv_k2_' + tg])
        # slab storage with a non-trivial row stride
        slab = t.zeros((Nz, 11, Nr), dtype=t.complex128, device='cuda')
        a = {k: slab[:, i, :] for i, k in enumerate(names)}

        def load():
            for k in names:
                a[k].copy_(dev(hip, g['sp_in_%s_m%d' % (k, m)]))
        load()
        rs = hip.row_stride(a['Jp'])
        hip.check(hip.lib().fb_correct_currents_curlfree_standard(
            p(a['rho_prev']), p(a['rho_next']), p(a['Jp']), p(a['Jm']), p(a['Jz']), rs,
            p(dkz), p(dkr), p(dik2), 1. / dt, Nz, Nr, hip.stream()), 'cc')
        for k in ('Jp', 'Jm', 'Jz'):
            assert rel_err(host(a[k]), g['cc_%s_m%d' % (k, m)]) < TOL
        tabs = [dev(hip, gs[k + '_' + tg]) for k in ('rho_prev_coef', 'rho_next_coef', 'j_coef', 'C', 'S_w')]
        for utr in (0, 1):
            load()
            hip.check(hip.lib().fb_push_eb_standard(
                *[p(a[k]) for k in names], rs, *[p(x) for x in tabs], p(dkr), p(dkz), dt, utr,
                c, epsilon_0, mu_0, Nz, Nr, hip.stream()), 'push_eb')
            for k in names[:6]:
                assert rel_err(host(a[k]), g['pe%d_%s_m%d' % (utr, k, m)]) < TOL, (utr, k)
        load()
        fz, fr = dev(hip, gs['filter_z_' + tg]), dev(hip, gs['filter_r_' + tg])
        fl = [a['Jp'], a['Jm'], a['Jz'], a['rho_next']]
        hip.check(hip.lib().fb_filter(4, hip.ptr_array(fl), rs, p(fz), p(fr), Nz, Nr, hip.stream()), 'filter')
        for k in ('Jp', 'Jm', 'Jz', 'rho_next'):
            assert rel_err(host(a[k]), g['fl_%s_m%d' % (k, m)]) < TOL
        # push_rho, erase, divide_by_volume, scale, rt<->pm
        load()
        hip.check(hip.lib().fb_push_rho(p(a['rho_prev']), p(a['rho_next']), rs, Nz, Nr, hip.stream()), 'pr')
        assert np.array_equal(host(a['rho_prev']), g['sp_in_rho_next_m%d' % m])
        assert np.all(host(a['rho_next']) == 0)
        load()
        inv = np.linspace(1., 2., Nr)
        hip.check(hip.lib().fb_divide_by_volume(2, hip.ptr_array([a['Jp'], a['Jm']]), rs,
                                                p(dev(hip, inv)), Nz, Nr, hip.stream()), 'div')
        assert np.array_equal(host(a['Jp']), g['sp_in_Jp_m%d' % m] * inv[None, :])
        r0, t0 = g['sp_in_Ep_m%d' % m], g['sp_in_Em_m%d' % m]
        pp, mm = np.empty_like(r0), np.empty_like(r0)
        oracle.rt_to_pm(r0.copy(), t0.copy(), pp, mm)
        hip.check(hip.lib().fb_rt_to_pm(1, hip.ptr_array([a['Ep']]), hip.ptr_array([a['Em']]),
                                        hip.ptr_array([a['Ep']]), hip.ptr_array([a['Em']]), rs,
                                        Nz, Nr, hip.stream()), 'rt2pm')
        assert np.array_equal(host(a['Ep']), pp) and np.array_equal(host(a['Em']), mm)
        hip.check(hip.lib().fb_pm_to_rt(1, hip.ptr_array([a['Ep']]), hip.ptr_array([a['Em']]),
                                        hip.ptr_array([a['Ep']]), hip.ptr_array([a['Em']]), rs,
                                        Nz, Nr, hip.stream()), 'pm2rt')
        rr, tt = np.empty_like(r0), np.empty_like(r0)
        oracle.pm_to_rt(pp, mm, rr, tt)
        assert np.array_equal(host(a['Ep']), rr) and np.array_equal(host(a['Em']), tt)
        hip.check(hip.lib().fb_erase(3, hip.ptr_array([a['Ep'], a['Em'], a['Ez']]), rs, Nz, Nr,
                                     hip.stream()), 'erase')
        assert np.all(host(slab[:, :3, :]) == 0) and np.any(host(slab[:, 3, :]) != 0)


# ------------------------------------------------------------------ FFT / Hankel
@pytest.mark.parametrize('Nz', [6, 48, 60, 62, 690, 2400, 4416, 1024, 384, 960, 2112, 4608])
def test_fft_generic_any_length(hip, Nz):
    """fb_fft_generic (pass-per-launch fallback for the lengths rocFFT refuses) against numpy:
    out of place and in place, odd and even pass counts, radices 2..31."""
    rng = np.random.default_rng(5)
    t = hip.torch()
    ncols = 37
    a = rng.normal(size=(Nz, ncols + 3)) + 1j * rng.normal(size=(Nz, ncols + 3))
    assert hip.lib().fb_fft_generic_supported(Nz)
    assert not hip.lib().fb_fft_generic_supported(37 * 2)
    src = dev(hip, a)
    dst = t.zeros((Nz, ncols + 5), dtype=t.complex128, device='cuda')
    scr = t.zeros((Nz, ncols + 1), dtype=t.complex128, device='cuda')
    call = hip.lib().fb_fft_generic
    hip.check(call(Nz, ncols, src.data_ptr(), ncols + 3, dst.data_ptr(), ncols + 5, scr.data_ptr(),
                   ncols + 1, -1, hip.stream()), 'generic fwd')
    assert rel_err(host(dst[:, :ncols]), np.fft.fft(a[:, :ncols], axis=0)) < TOL
    assert np.all(host(dst[:, ncols:]) == 0)
    hip.check(call(Nz, ncols, src.data_ptr(), ncols + 3, src.data_ptr(), ncols + 3, scr.data_ptr(),
                   ncols + 1, +1, hip.stream()), 'generic bwd in place')
    assert rel_err(host(src[:, :ncols]), np.fft.ifft(a[:, :ncols], axis=0)) < TOL
    assert np.array_equal(host(src[:, ncols:]), a[:, ncols:])



@pytest.mark.parametrize('Nz,Nr,nf', [(32, 16, 1), (200, 64, 3), (254, 50, 2), (1024, 128, 6),
                                      (64, 24, 2), (128, 33, 3), (256, 64, 2), (512, 20, 1),
                                      (2048, 16, 3), (4096, 9, 2), (576, 24, 2), (1152, 128, 3),
                                      (2304, 10, 1), (4416, 6, 2), (4288, 5, 2), (4119, 3, 1)])
def test_fft_matches_numpy(hip, Nz, Nr, nf):
    """Both z-FFT paths (hand-written kernel for Nz = 2^k in [64, 4096] and 9 * 2^k, rocFFT for
    the rest) against numpy, on a strided sub-view of a slab; ragged column counts included."""
    from fbpic_amd.fields.spectral_transform.fourier import fft_exec
    rng = np.random.default_rng(3)
    t = hip.torch()
    a = rng.normal(size=(Nz, nf + 1, Nr)) + 1j * rng.normal(size=(Nz, nf + 1, Nr))
    src = dev(hip, a)
    dst = t.zeros((Nz, nf + 2, Nr), dtype=t.complex128, device='cuda')
    fft_exec(src[:, 1, :], dst[:, 2, :], -1, ncols=nf * Nr)
    ref = np.fft.fft(a[:, 1:, :], axis=0)
    assert rel_err(host(dst[:, 2:, :]), ref) < TOL
    assert np.all(host(dst[:, :2, :]) == 0)           # neighbours untouched
    fft_exec(src[:, 1, :], dst[:, 2, :], +1, ncols=nf * Nr)
    assert rel_err(host(dst[:, 2:, :]), np.fft.ifft(a[:, 1:, :], axis=0)) < TOL
    # in place, and the two paths against each other where both exist
    fft_exec(src[:, 1, :], src[:, 1, :], -1, ncols=nf * Nr)
    assert rel_err(host(src[:, 1:, :]), ref) < TOL
    assert np.array_equal(host(src[:, 0, :]), a[:, 0, :])
    if hip.lib().fb_zfft_supported(Nz):
        from fbpic_amd.fields.spectral_transform import fourier
        src2 = dev(hip, a)
        fourier.USE_ZFFT = False
        try:
            fft_exec(src2[:, 1, :], dst[:, 2, :], -1, ncols=nf * Nr)
        finally:
            fourier.USE_ZFFT = True
        assert rel_err(host(dst[:, 2:, :]), ref) < TOL


@pytest.mark.parametrize('Nz,Nr', [(32, 16), (100, 50), (200, 64), (1024, 128), (70, 130)])
def test_hankel_gemm(hip, Nz, Nr):
    """fp64 MFMA GEMM with an ASYMMETRIC matrix (catches row/column fragment swaps)."""
    rng = np.random.default_rng(4)
    t = hip.torch()
    njobs = 3
    a = rng.normal(size=(Nz, njobs, Nr)) + 1j * rng.normal(size=(Nz, njobs, Nr))
    mats = [rng.normal(size=(Nr, Nr)) * np.exp(rng.uniform(-5, 5, size=(Nr, 1))) for _ in range(njobs)]
    src = dev(hip, a)
    dst = t.zeros((Nz, njobs + 1, Nr), dtype=t.complex128, device='cuda')
    dm = [dev(hip, m) for m in mats]
    ins = [src[:, j, :] for j in range(njobs)]
    outs = [dst[:, j, :] for j in range(njobs)]
    hip.check(hip.lib().fb_hankel(njobs, hip.ptr_array(ins), njobs * Nr, hip.ptr_array(outs),
                                  (njobs + 1) * Nr, hip.ptr_array(dm), 0.5, Nz, Nr, hip.stream()), 'hk')
    for j in range(njobs):
        ref = 0.5 * (a[:, j, :] @ mats[j])
        # error bound relative to sum |a||m| (ill-scaled rows): compare to the fp64 product
        scale = (np.abs(a[:, j, :]) @ np.abs(mats[j])).max()
        assert np.abs(host(dst[:, j, :]) - ref).max() < 1e-14 * scale
    assert np.all(host(dst[:, njobs, :]) == 0)


def test_transformer_vs_golden(hip):
    from fbpic_amd.fields.spectral_transform.spectral_transformer import SpectralTransformer
    g = golden('spectral')
    Nz, Nr, Nm = int(g['Nz']), int(g['Nr']), int(g['Nm'])
    t = hip.torch()
    for m in range(Nm):
        tr = SpectralTransformer(Nz, Nr, m, Nr * float(g['dr']))
        a, r, tt = (dev(hip, g[k + '_m%d' % m]) for k in ('in_scal', 'in_r', 'in_t'))
        o1 = t.empty_like(a); o2 = t.empty_like(a)
        tr.interp2spect_scal(a, o1)
        achieved('transformer i2s_scal m=%d' % m, rel_err(host(o1), g['i2s_scal_m%d' % m]), TOL_TRANSFORM)
        tr.spect2interp_scal(a, o1)
        achieved('transformer s2i_scal m=%d' % m, rel_err(host(o1), g['s2i_scal_m%d' % m]), TOL_TRANSFORM)
        tr.interp2spect_vect(r, tt, o1, o2)
        achieved('transformer i2s_p m=%d' % m, rel_err(host(o1), g['i2s_p_m%d' % m]), TOL_TRANSFORM)
        achieved('transformer i2s_m m=%d' % m, rel_err(host(o2), g['i2s_m_m%d' % m]), TOL_TRANSFORM)
        tr.spect2interp_vect(r, tt, o1, o2)
        achieved('transformer s2i_r m=%d' % m, rel_err(host(o1), g['s2i_r_m%d' % m]), TOL_TRANSFORM)
        achieved('transformer s2i_t m=%d' % m, rel_err(host(o2), g['s2i_t_m%d' % m]), TOL_TRANSFORM)


def test_hankel_scaled_fusions(hip):
    """divide-by-volume (input column scale) and filter (output row/column scale) fused
    into the GEMM == the three separate passes."""
    rng = np.random.default_rng(8)
    Nz, Nr, njobs = 96, 72, 4
    t = hip.torch()
    a = rng.normal(size=(Nz, njobs, Nr)) + 1j * rng.normal(size=(Nz, njobs, Nr))
    mats = [rng.normal(size=(Nr, Nr)) for _ in range(njobs)]
    sk = [rng.uniform(0.5, 2., Nr), None, rng.uniform(0.5, 2., Nr), None]
    fz = [rng.uniform(0., 1., Nz), rng.uniform(0., 1., Nz), None, None]
    fr = [rng.uniform(0., 1., Nr), rng.uniform(0., 1., Nr), None, None]
    src = dev(hip, a)
    dst = t.zeros((Nz, njobs, Nr), dtype=t.complex128, device='cuda')
    D = lambda L: [dev(hip, v) if v is not None else None for v in L]  # noqa: E731
    dm, dsk, dfz, dfr = D(mats), D(sk), D(fz), D(fr)
    pa = hip.ptr_array
    hip.check(hip.lib().fb_hankel_scaled(njobs, pa([src[:, j, :] for j in range(njobs)]), njobs * Nr,
                                         pa([dst[:, j, :] for j in range(njobs)]), njobs * Nr,
                                         pa(dm), pa(dsk), pa(dfz), pa(dfr), 1.0, Nz, Nr,
                                         hip.stream()), 'hks')
    for j in range(njobs):
        x = a[:, j, :] * (sk[j][None, :] if sk[j] is not None else 1.)
        ref = x @ mats[j]
        if fz[j] is not None:
            ref = (fz[j][:, None] * fr[j][None, :]) * ref
        scale = (np.abs(a[:, j, :]) @ np.abs(mats[j])).max() * 2
        assert np.abs(host(dst[:, j, :]) - ref).max() < 1e-14 * scale, j


def test_rt_pm_fused_into_hankel_and_fft(hip):
    """fb_hankel_rt_to_pm_scaled == fb_rt_to_pm then fb_hankel_scaled (bit-identical: same
    operand values, same MFMA order); fb_zfft_pm_to_rt == fb_pm_to_rt then fb_zfft backward."""
    import ctypes
    rng = np.random.default_rng(9)
    Nz, Nr = 128, 48
    t = hip.torch()
    pa = hip.ptr_array
    nf = 7                                   # two vector triples + one scalar
    a = rng.normal(size=(Nz, nf, Nr)) + 1j * rng.normal(size=(Nz, nf, Nr))
    mats = [dev(hip, rng.normal(size=(Nr, Nr))) for _ in range(nf)]
    sk = [dev(hip, rng.uniform(0.5, 2., Nr)) for _ in range(nf)]
    fz = [dev(hip, rng.uniform(0., 1., Nz)) for _ in range(nf)]
    fr = [dev(hip, rng.uniform(0., 1., Nr)) for _ in range(nf)]
    # reference sequence
    s1 = dev(hip, a)
    f1 = [s1[:, j, :] for j in range(nf)]
    r_, t_ = pa(f1[0:6:3]), pa(f1[1:6:3])
    hip.check(hip.lib().fb_rt_to_pm(2, r_, t_, r_, t_, nf * Nr, Nz, Nr, hip.stream()), 'rt_to_pm')
    d1 = t.zeros((Nz, nf, Nr), dtype=t.complex128, device='cuda')
    hip.check(hip.lib().fb_hankel_scaled(nf, pa(f1), nf * Nr, pa([d1[:, j, :] for j in range(nf)]),
                                         nf * Nr, pa(mats), pa(sk), pa(fz), pa(fr), 1.0, Nz, Nr,
                                         hip.stream()), 'hks')
    # fused
    s2 = dev(hip, a)
    f2 = [s2[:, j, :] for j in range(nf)]
    ins = [f2[0], f2[0], f2[2], f2[3], f2[3], f2[5], f2[6]]
    in2 = [f2[1], f2[1], None, f2[4], f2[4], None, None]
    sgn = (ctypes.c_double * nf)(-1., 1., 0., -1., 1., 0., 0.)
    d2 = t.zeros((Nz, nf, Nr), dtype=t.complex128, device='cuda')
    hip.check(hip.lib().fb_hankel_rt_to_pm_scaled(
        nf, pa(ins), pa(in2), sgn, nf * Nr, pa([d2[:, j, :] for j in range(nf)]), nf * Nr, pa(mats),
        pa(sk), pa(fz), pa(fr), 1.0, Nz, Nr, hip.stream()), 'hk rt')
    assert np.array_equal(host(d1), host(d2))
    assert np.array_equal(host(s2), a)                      # inputs untouched
    # backward FFT with (p, m) -> (r, t) on load
    b = rng.normal(size=(Nz, 6, Nr)) + 1j * rng.normal(size=(Nz, 6, Nr))
    s3 = dev(hip, b)
    f3 = [s3[:, j, :] for j in range(6)]
    p_, m_ = pa(f3[0::3]), pa(f3[1::3])
    hip.check(hip.lib().fb_pm_to_rt(2, p_, m_, p_, m_, 6 * Nr, Nz, Nr, hip.stream()), 'pm_to_rt')
    o3 = t.zeros((Nz, 7, Nr), dtype=t.complex128, device='cuda')
    hip.check(hip.lib().fb_zfft(Nz, 6 * Nr, s3.data_ptr(), 6 * Nr, o3[:, 1, :].data_ptr(), 7 * Nr, +1,
                                hip.stream()), 'zfft')
    s4 = dev(hip, b)
    o4 = t.zeros((Nz, 7, Nr), dtype=t.complex128, device='cuda')
    hip.check(hip.lib().fb_zfft_pm_to_rt(Nz, 6 * Nr, s4.data_ptr(), 6 * Nr, o4[:, 1, :].data_ptr(),
                                         7 * Nr, Nr, hip.stream()), 'zfft pm')
    assert np.array_equal(host(o3), host(o4))
    assert np.array_equal(host(s4), b)


def test_zfft_from_records(hip):
    """fb_zfft_from_records == fb_zfft forward of the same fields laid out as a slab."""
    rng = np.random.default_rng(10)
    Nz, Nr, nf, rec = 256, 24, 5, 8
    t = hip.torch()
    a = rng.normal(size=(Nz, Nr, rec)) + 1j * rng.normal(size=(Nz, Nr, rec))
    S = dev(hip, a)
    out = t.zeros((Nz, nf + 1, Nr), dtype=t.complex128, device='cuda')
    hip.check(hip.lib().fb_zfft_from_records(Nz, nf, Nr, S.data_ptr(), Nr * rec, rec,
                                             out[:, 0, :].data_ptr(), (nf + 1) * Nr, hip.stream()),
              'zfft records')
    ref = np.fft.fft(np.transpose(a[:, :, :nf], (0, 2, 1)), axis=0)
    assert rel_err(host(out[:, :nf, :]), ref) < TOL
    assert np.all(host(out[:, nf, :]) == 0)


@pytest.mark.parametrize('Nz', [4416, 960])
def test_fft_generic_from_records_consume(hip, Nz):
    """fb_fft_generic_from_records_consume (two-sweep lengths 192 x R): forward transform of the
    whole record array == numpy, and the records are zero afterwards."""
    rng = np.random.default_rng(12)
    Nr, rec = 24, 8
    t = hip.torch()
    assert hip.lib().fb_fft_generic_from_records_supported(Nz)
    assert not hip.lib().fb_fft_generic_from_records_supported(1024)
    a = rng.normal(size=(Nz, Nr, rec)) + 1j * rng.normal(size=(Nz, Nr, rec))
    S = dev(hip, a)
    out = t.zeros((Nz, rec + 1, Nr), dtype=t.complex128, device='cuda')
    scr = t.zeros((Nz, rec * Nr + 8), dtype=t.complex128, device='cuda')
    hip.check(hip.lib().fb_fft_generic_from_records_consume(
        Nz, rec, Nr, S.data_ptr(), Nr * rec, rec, out[:, 0, :].data_ptr(), (rec + 1) * Nr,
        scr.data_ptr(), rec * Nr + 8, hip.stream()), 'generic records')
    ref = np.fft.fft(np.transpose(a, (0, 2, 1)), axis=0)
    assert rel_err(host(out[:, :rec, :]), ref) < TOL
    assert np.all(host(out[:, rec, :]) == 0)
    assert np.all(host(S) == 0)


def test_psatd_step_fused_equals_separate(hip, oracle):
    """fb_psatd_step_standard == correct_currents -> push_eb -> push_rho (oracle, per mode)."""
    g = golden('spectral')
    gs = golden('grid_setup')
    Nm, Nz, Nr = int(g['Nm']), int(g['Nz']), int(g['Nr'])
    dt = float(g['dt'])
    names = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']
    t = hip.torch()
    for correct in (1, 0, 2):       # 2 = correction only (decomposed-domain step)
        for utr in (0, 1):
            slab = t.zeros((Nz, 11 * Nm, Nr), dtype=t.complex128, device='cuda')
            fields, tables, expect = [], [], []
            keep = []
            for m in range(Nm):
                tg = 'o-1_m%d' % m
                kz = np.repeat(gs['kz_' + tg][:, None], Nr, 1).copy()
                kr = np.repeat(gs['kr_' + tg][None, :], Nz, 0).copy()
                a = {k: g['sp_in_%s_m%d' % (k, m)].copy() for k in names}
                for i, k in enumerate(names):
                    slab[:, 11 * m + i, :] = dev(hip, a[k])
                    fields.append(slab[:, 11 * m + i, :])
                tabs = [gs[k + '_' + tg] for k in ('rho_prev_coef', 'rho_next_coef', 'j_coef', 'C', 'S_w')]
                tabs += [kr, kz, gs['inv_k2_' + tg]]
                dt_ = [dev(hip, x) for x in tabs]
                keep.append(dt_)
                tables += dt_
                if correct:
                    oracle.correct_currents_curlfree(a['rho_prev'], a['rho_next'], a['Jp'], a['Jm'],
                                                     a['Jz'], kz, kr, gs['inv_k2_' + tg], 1. / dt)
                if correct != 2:
                    oracle.push_eb_standard(*[a[k] for k in names], *tabs[:5], kr, kz, dt, utr)
                    a['rho_prev'] = a['rho_next'].copy()
                    a['rho_next'][:] = 0.
                expect.append(a)
            hip.check(hip.lib().fb_psatd_step_standard(Nm, hip.ptr_array(fields), 11 * Nm * Nr,
                                                       hip.ptr_array(tables), dt, correct, utr, c,
                                                       epsilon_0, mu_0, Nz, Nr, hip.stream()), 'psatd')
            for m in range(Nm):
                for i, k in enumerate(names):
                    assert rel_err(host(slab[:, 11 * m + i, :]), expect[m][k]) < TOL or \
                        (np.abs(expect[m][k]).max() == 0 and np.all(host(slab[:, 11 * m + i, :]) == 0)), \
                        (correct, utr, m, k)


@pytest.mark.parametrize('tag', ['gal', 'com', 'gal0'])
def test_comoving_spectral_kernels_vs_golden(hip, tag):
    """fb_correct_currents_curlfree_comoving / fb_push_eb_comoving against the reference's
    numba kernels (Galilean V = 0.9 c, comoving V = -0.5 c, Galilean V = 0), on slab views."""
    g = golden('galilean_kernels')
    Nz, Nr, Nm = int(g['Nz']), int(g['Nr']), int(g['Nm'])
    V, dt = float(g[tag + '_V']), float(g['dt'])
    names = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']
    t = hip.torch()
    p = hip.ptr
    for m in range(Nm):
        tb = {k: dev(hip, np.ascontiguousarray(g['%s_%s_m%d' % (tag, k, m)], dtype=np.complex128))
              for k in ('j_coef', 'rho_prev_coef', 'rho_next_coef', 'T_eb', 'T_cc', 'T_rho', 'j_corr_coef')}
        tr = {k: dev(hip, np.ascontiguousarray(g['%s_%s_m%d' % (tag, k, m)], dtype=np.float64))
              for k in ('C', 'S_w', 'kz', 'kr', 'inv_k2')}

        def slab_with_inputs():
            slab = t.zeros((Nz, 12, Nr), dtype=t.complex128, device='cuda')
            for i, k in enumerate(names):
                slab[:, i, :] = dev(hip, g['%s_in_%s_m%d' % (tag, k, m)])
            return slab, {k: slab[:, i, :] for i, k in enumerate(names)}
        slab, a = slab_with_inputs()
        hip.check(hip.lib().fb_correct_currents_curlfree_comoving(
            p(a['rho_prev']), p(a['rho_next']), p(a['Jp']), p(a['Jm']), p(a['Jz']), 12 * Nr,
            p(tr['kz']), p(tr['kr']), p(tr['inv_k2']), p(tb['j_corr_coef']), p(tb['T_eb']),
            p(tb['T_cc']), Nz, Nr, hip.stream()), 'cc comoving')
        for k in ('Jp', 'Jm', 'Jz'):
            assert rel_err(host(a[k]), g['%s_cc_%s_m%d' % (tag, k, m)]) < TOL, (m, k)
        for utr in (0, 1):
            slab, a = slab_with_inputs()
            hip.check(hip.lib().fb_push_eb_comoving(
                *[p(a[k]) for k in names], 12 * Nr, p(tb['rho_prev_coef']), p(tb['rho_next_coef']),
                p(tb['j_coef']), p(tr['C']), p(tr['S_w']), p(tb['T_eb']), p(tb['T_cc']), p(tb['T_rho']),
                p(tr['kr']), p(tr['kz']), dt, V, utr, c, epsilon_0, mu_0, Nz, Nr, hip.stream()),
                'push comoving')
            for k in names[:6]:
                assert rel_err(host(a[k]), g['%s_pe%d_%s_m%d' % (tag, utr, k, m)]) < TOL, (m, utr, k)
            assert np.all(host(slab[:, 11, :]) == 0)


@pytest.mark.parametrize('tag', ['std', 'gal', 'com'])
def test_crossdeposition_kernels_vs_golden(hip, tag):
    """fb_correct_currents_crossdeposition_{standard,comoving} against the reference's numba
    kernels (numba_methods.py:87-116, 243-275), on slab views."""
    g = golden('crossdep_kernels')
    Nz, Nr, Nm = int(g['Nz']), int(g['Nr']), int(g['Nm'])
    names = ['rho_prev', 'rho_next', 'rho_next_z', 'rho_next_xy', 'Jp', 'Jm', 'Jz']
    t = hip.torch()
    p = hip.ptr
    for m in range(Nm):
        kz, kr = (dev(hip, np.ascontiguousarray(g['%s_%s_m%d' % (tag, k, m)], dtype=np.float64))
                  for k in ('kz', 'kr'))
        slab = t.zeros((Nz, 8, Nr), dtype=t.complex128, device='cuda')
        for i, k in enumerate(names):
            slab[:, i, :] = dev(hip, g['%s_in_%s_m%d' % (tag, k, m)])
        a = {k: slab[:, i, :] for i, k in enumerate(names)}
        if tag == 'std':
            hip.check(hip.lib().fb_correct_currents_crossdeposition_standard(
                *[p(a[k]) for k in names], 8 * Nr, p(kz), p(kr), 1. / float(g['dt']), Nz, Nr,
                hip.stream()), 'cross std')
        else:
            tb = {k: dev(hip, np.ascontiguousarray(g['%s_%s_m%d' % (tag, k, m)], dtype=np.complex128))
                  for k in ('j_corr_coef', 'T_eb', 'T_cc')}
            hip.check(hip.lib().fb_correct_currents_crossdeposition_comoving(
                *[p(a[k]) for k in names], 8 * Nr, p(kz), p(kr), p(tb['j_corr_coef']),
                p(tb['T_eb']), p(tb['T_cc']), Nz, Nr, hip.stream()), 'cross comoving')
        for k in ('Jp', 'Jm', 'Jz'):
            assert rel_err(host(a[k]), g['%s_cc_%s_m%d' % (tag, k, m)]) < TOL, (m, k)
        for i, k in enumerate(names[:4]):       # the charge densities are read-only
            assert np.array_equal(host(a[k]), g['%s_in_%s_m%d' % (tag, k, m)]), k
        assert np.all(host(slab[:, 7, :]) == 0)


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_gather_push_fused_is_bit_identical_to_sequence(hip, shape):
    """fb_gather_push == fb_gather -> fb_push_p -> fb_push_x (same arithmetic, same bits)."""
    g = golden('gather')
    Nz, Nr, nm = int(g['Nz']), int(g['Nr']), 3
    n = g['x'].size
    t = hip.torch()
    rng = np.random.default_rng(21)
    views = [dev(hip, g['grids'][m, k] * 1e9) for m in range(nm) for k in range(6)]
    u0 = [rng.normal(size=n) for _ in range(3)]
    ig0 = 1. / np.sqrt(1 + u0[0]**2 + u0[1]**2 + u0[2]**2)
    dt = 6.67e-16
    geom = (float(g['rmax_gather']), 1. / float(g['dz']), float(g['zmin']), Nz, 1. / float(g['dr']), 0., Nr)
    p = hip.ptr
    sh = 1 if shape == 'linear' else 3

    def fresh():
        pos = [dev(hip, g[k]) for k in ('x', 'y', 'z')]
        mom = [dev(hip, a) for a in u0] + [dev(hip, ig0)]
        F = [t.zeros(n, dtype=t.float64, device='cuda') for _ in range(6)]
        return pos, mom, F
    pos, mom, F = fresh()
    hip.check(hip.lib().fb_gather(sh, nm, n, *[p(a) for a in pos], *geom, hip.ptr_array(views), Nr,
                                  *[p(f) for f in F], hip.stream()), 'gather')
    hip.check(hip.lib().fb_push_p(n, *[p(a) for a in mom], *[p(f) for f in F], -e, m_e, c, dt,
                                  hip.stream()), 'push_p')
    hip.check(hip.lib().fb_push_x(n, *[p(a) for a in pos], *[p(a) for a in mom], c, 0.5 * dt,
                                  1., 1., 1., hip.stream()), 'push_x')
    pos2, mom2, F2 = fresh()
    hip.check(hip.lib().fb_gather_push(sh, nm, n, *[p(a) for a in pos2], *[p(a) for a in mom2], *geom,
                                       hip.ptr_array(views), Nr, *[p(f) for f in F2], -e, m_e, c, dt,
                                       0.5 * dt, 0., 0., hip.stream()), 'gather_push')
    for a, b in zip(pos + mom + F, pos2 + mom2 + F2):
        assert np.array_equal(host(a), host(b))
    # with the periodic wrap folded in == fb_shift_periodic first, then the same call
    zlo = float(g['z'].min()) + 0.2 * float(g['z'].max() - g['z'].min())
    zhi = float(g['z'].max()) - 0.2 * float(g['z'].max() - g['z'].min())
    pos4, mom4, F4 = fresh()
    hip.check(hip.lib().fb_shift_periodic(n, p(pos4[2]), zlo, zhi, hip.stream()), 'shift')
    hip.check(hip.lib().fb_gather_push(sh, nm, n, *[p(a) for a in pos4], *[p(a) for a in mom4], *geom,
                                       hip.ptr_array(views), Nr, *[p(f) for f in F4], -e, m_e, c, dt,
                                       0.5 * dt, 0., 0., hip.stream()), 'gather_push')
    pos5, mom5, F5 = fresh()
    hip.check(hip.lib().fb_gather_push(sh, nm, n, *[p(a) for a in pos5], *[p(a) for a in mom5], *geom,
                                       hip.ptr_array(views), Nr, *[p(f) for f in F5], -e, m_e, c, dt,
                                       0.5 * dt, zlo, zhi, hip.stream()), 'gather_push wrap')
    for a, b in zip(pos4 + mom4 + F4, pos5 + mom5 + F5):
        assert np.array_equal(host(a), host(b))
    # fields not stored, no position push
    pos3, mom3, _ = fresh()
    hip.check(hip.lib().fb_gather_push(sh, nm, n, *[p(a) for a in pos3], *[p(a) for a in mom3], *geom,
                                       hip.ptr_array(views), Nr, *([None] * 6), -e, m_e, c, dt,
                                       0., 0., 0., hip.stream()), 'gather_push')
    for a, b in zip(mom, mom3):
        assert np.array_equal(host(a), host(b))
    for k, b in zip(('x', 'y', 'z'), pos3):
        assert np.array_equal(g[k], host(b))


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_gather_push_rank_next(hip, oracle, shape):
    """fb_gather_push_rank_next == fb_gather_push (bit-identical particle arrays) + the cell /
    rank by-product for the NEXT push_x: followed by fb_push_x_bin_sort_particles(preranked = 1)
    it gives the same sorted arrays, cells and prefix sums as the un-preranked sort of the
    fb_gather_push result (cells bit-identical to the oracle's index of the oracle-pushed
    positions)."""
    g = golden('gather')
    Nz, Nr, nm = int(g['Nz']), int(g['Nr']), 2
    n = g['x'].size
    t = hip.torch()
    rng = np.random.default_rng(22)
    views = [dev(hip, g['grids'][m, k] * 1e9) for m in range(nm) for k in range(6)]
    u0 = [rng.normal(size=n) for _ in range(3)]
    ig0 = 1. / np.sqrt(1 + u0[0]**2 + u0[1]**2 + u0[2]**2)
    w0 = rng.uniform(0.5, 1.5, n)
    dt = 6.67e-16
    invdz, zmin, invdr = 1. / float(g['dz']), float(g['zmin']), 1. / float(g['dr'])
    geom = (float(g['rmax_gather']), invdz, zmin, Nz, invdr, 0., Nr)
    cgeom = (invdz, zmin, Nz, invdr, 0., Nr)
    p = hip.ptr
    sh = 1 if shape == 'linear' else 3
    ncell = Nz * (Nr + 1)
    nb = int(hip.lib().fb_bin_sort_workspace_bytes(n, ncell))

    def fresh():
        pos = [dev(hip, g[k]) for k in ('x', 'y', 'z')]
        mom = [dev(hip, a) for a in u0] + [dev(hip, ig0)]
        return pos, mom
    posA, momA = fresh()
    hip.check(hip.lib().fb_gather_push(sh, nm, n, *[p(a) for a in posA], *[p(a) for a in momA], *geom,
                                       hip.ptr_array(views), Nr, *([None] * 6), -e, m_e, c, dt,
                                       0.5 * dt, 0., 0., hip.stream()), 'gather_push')
    posB, momB = fresh()
    ws = t.empty(nb, dtype=t.uint8, device='cuda')
    hip.check(hip.lib().fb_gather_push_rank_next(
        sh, nm, n, *[p(a) for a in posB], *[p(a) for a in momB], *geom, hip.ptr_array(views), Nr,
        *([None] * 6), -e, m_e, c, dt, 0.5 * dt, 0., 0., 0.5 * dt, 1., 1., 1., ncell, p(ws), nb, 0,
        hip.stream()), 'gather_push_rank_next')
    for a, b in zip(posA + momA, posB + momB):
        assert np.array_equal(host(a), host(b))
    wdev = dev(hip, w0)

    def sort(pos, mom, work, preranked):
        src = [pos[0], pos[1], pos[2], mom[0], mom[1], mom[2], wdev, mom[3]]
        dst = [t.empty_like(a) for a in src]
        ci = t.empty(n, dtype=t.int32, device='cuda')
        si = t.empty(n, dtype=t.int32, device='cuda')
        pre = t.empty(ncell, dtype=t.int32, device='cuda')
        hip.check(hip.lib().fb_push_x_bin_sort_particles(
            n, ncell, p(src[0]), p(src[1]), p(src[2]), p(src[3]), p(src[4]), p(src[5]), p(src[7]),
            c, 0.5 * dt, 1., 1., 1., *cgeom, 8, hip.ptr_array(src), hip.ptr_array(dst), p(ci), p(si),
            p(pre), p(work), nb, preranked, hip.stream()), 'push_x_bin_sort')
        return [host(a) for a in dst], host(ci), host(si), host(pre)
    dA, ciA, siA, preA = sort(posA, momA, t.empty(nb, dtype=t.uint8, device='cuda'), 0)
    dB, ciB, siB, preB = sort(posB, momB, ws, 1)
    assert np.array_equal(preA, preB) and np.array_equal(ciA, ciB)
    assert np.array_equal(np.sort(siB), np.arange(n, dtype=np.int32))
    # same multiset of particles per cell (the order inside a cell is free): compare after a
    # canonical sort by (cell, x, y, z)
    oA = np.lexsort((dA[2], dA[1], dA[0], ciA))
    oB = np.lexsort((dB[2], dB[1], dB[0], ciB))
    for a, b in zip(dA, dB):
        assert np.array_equal(a[oA], b[oB])
    # cells of the twice-pushed positions, oracle arithmetic
    xr, yr, zr = (host(a).copy() for a in posA)
    oracle.push_x(xr, yr, zr, host(momA[0]), host(momA[1]), host(momA[2]), host(momA[3]), 0.5 * dt)
    ref = oracle.cell_index(xr, yr, zr, *cgeom)
    assert np.array_equal(ciB, ref[siB])


@pytest.mark.parametrize('lo,hi', [(0, 0), (300, 901), (37, 38), (64, 1536), (1, 1535), (999, 1536)])
def test_gather_push_rank_next_range(hip, lo, hi):
    """fb_gather_push_rank_next_range: the pass over [lo, hi) followed by the pass over everything
    else (bounds read on the device, not aligned to the 64-particle chunks of the kernel, so a
    wave's active lanes are a prefix, a suffix or both) == one pass over all particles: the
    particle arrays bit-identical, the same cell for every particle, and ranks that are a
    permutation of 0 .. count-1 inside every cell (what the counting sort needs)."""
    g = golden('gather')
    Nz, Nr, nm = int(g['Nz']), int(g['Nr']), 2
    n = g['x'].size
    assert n >= hi
    t = hip.torch()
    rng = np.random.default_rng(23)
    views = [dev(hip, g['grids'][m, k] * 1e9) for m in range(nm) for k in range(6)]
    u0 = [rng.normal(size=n) for _ in range(3)]
    ig0 = 1. / np.sqrt(1 + u0[0]**2 + u0[1]**2 + u0[2]**2)
    dt = 6.67e-16
    invdz, zmin, invdr = 1. / float(g['dz']), float(g['zmin']), 1. / float(g['dr'])
    geom = (float(g['rmax_gather']), invdz, zmin, Nz, invdr, 0., Nr)
    p = hip.ptr
    ncell = Nz * (Nr + 1)
    nb = int(hip.lib().fb_bin_sort_workspace_bytes(n, ncell))
    bounds = t.tensor([lo, hi], dtype=t.int32, device='cuda')

    def run(split):
        pos = [dev(hip, g[k]) for k in ('x', 'y', 'z')]
        mom = [dev(hip, a) for a in u0] + [dev(hip, ig0)]
        eb = [t.zeros(n, dtype=t.float64, device='cuda') for _ in range(6)]
        ws = t.zeros(nb, dtype=t.uint8, device='cuda')
        passes = [(None, None, 0, 0)] if not split else \
            [(p(bounds[0:1]), p(bounds[1:2]), 1, 0), (p(bounds[0:1]), p(bounds[1:2]), 2, 1)]
        for lo_p, hi_p, mode, clean in passes:
            hip.check(hip.lib().fb_gather_push_rank_next_range(
                1, nm, n, *[p(a) for a in pos], *[p(a) for a in mom], *geom, hip.ptr_array(views), Nr,
                *[p(a) for a in eb], -e, m_e, c, dt, 0.5 * dt, 0., 0., 0.5 * dt, 1., 1., 1., ncell,
                p(ws), nb, clean, lo_p, hi_p, mode, hip.stream()), 'gather_push_rank_next_range')
        t.cuda.synchronize()
        w = ws.cpu().numpy()
        al = lambda v: (v + 255) // 256 * 256
        count = w[:4 * ncell].view(np.int32)
        cell = w[al(4 * ncell):al(4 * ncell) + 4 * n].view(np.int32)
        rank = w[al(4 * ncell) + al(4 * n):al(4 * ncell) + al(4 * n) + 4 * n].view(np.int32)
        return [host(a) for a in pos + mom + eb], count.copy(), cell.copy(), rank.copy()
    A, cntA, cellA, rankA = run(False)
    B, cntB, cellB, rankB = run(True)
    for a, b in zip(A, B):
        assert np.array_equal(a, b)
    assert np.array_equal(cellA, cellB) and np.array_equal(cntA, cntB)
    assert cntB.sum() == n and np.array_equal(np.bincount(cellB, minlength=ncell), cntB)
    for rank, cell in ((rankA, cellA), (rankB, cellB)):
        o = np.lexsort((rank, cell))
        first = np.concatenate(([0], np.cumsum(np.bincount(cell, minlength=ncell))[:-1]))
        assert np.array_equal(rank[o], np.arange(n) - first[cell[o]])


@pytest.mark.parametrize('presorted', [True, False])
def test_bin_sort_particles(hip, oracle, presorted):
    """Counting-sort fast path: cells sorted, a valid permutation, prefix sums and cell
    indices identical to the reference definition (order inside a cell is free)."""
    rng = np.random.default_rng(31)
    n, Nz, Nr = 300001, 96, 40
    dzc = 0.2e-6
    r = rng.uniform(0, 1.05 * Nr * dzc, n)
    th = rng.uniform(0, 2 * np.pi, n)
    x, y = r * np.cos(th), r * np.sin(th)
    z = rng.uniform(0, Nz * dzc, n)
    geom = (1. / dzc, 0., Nz, 1. / dzc, 0., Nr)
    ref = oracle.cell_index(x, y, z, *geom)
    if presorted:       # the PIC-cycle situation: almost sorted, long runs
        o = np.argsort(ref, kind='stable')
        x, y, z, ref = x[o], y[o], z[o], ref[o]
        sw = rng.integers(0, n - 40, 2000)
        for a in sw:     # perturb: a few particles out of place
            for arr in (x, y, z, ref):
                arr[a], arr[a + 37] = arr[a + 37], arr[a]
    w = rng.normal(size=n)
    t = hip.torch()
    ncell = Nz * (Nr + 1)
    src = [dev(hip, a) for a in (x, y, z, w)]
    dst = [t.empty_like(src[0]) for _ in range(4)]
    ci = t.empty(n, dtype=t.int32, device='cuda')
    si = t.empty(n, dtype=t.int32, device='cuda')
    pre = t.empty(ncell, dtype=t.int32, device='cuda')
    nb = int(hip.lib().fb_bin_sort_workspace_bytes(n, ncell))
    ws = t.empty(nb, dtype=t.uint8, device='cuda')
    p = hip.ptr
    for _ in range(2):     # twice: the per-cell counters must be reset between calls
        hip.check(hip.lib().fb_bin_sort_particles(n, ncell, p(src[0]), p(src[1]), p(src[2]), *geom, 4,
                                                  hip.ptr_array(src), hip.ptr_array(dst), p(ci), p(si),
                                                  p(pre), p(ws), nb, hip.stream()), 'binsort')
    cis, sidx, prefix = host(ci), host(si), host(pre)
    assert np.all(np.diff(cis) >= 0)
    assert np.array_equal(np.sort(sidx), np.arange(n, dtype=np.int32))
    assert np.array_equal(cis, ref[sidx])                       # bit-exact cell of every particle
    assert np.array_equal(prefix, np.cumsum(np.bincount(ref, minlength=ncell)).astype(np.int32))
    for a, b in zip((x, y, z, w), dst):
        assert np.array_equal(host(b), a[sidx])


@pytest.mark.parametrize('preranked', [0, 1])
def test_push_x_folded_into_sort(hip, oracle, preranked):
    """fb_push_x_bin_sort_particles == fb_push_x followed by the counting sort: pushed
    positions bit-identical to the oracle push, cells from the pushed positions.  With
    `preranked` the cell / rank pass is the by-product of fb_deposit_J_rank_next, whose J must
    equal fb_deposit_J's."""
    from scipy.constants import c
    rng = np.random.default_rng(32)
    n, Nz, Nr = 200003, 64, 32
    dzc = 0.2e-6
    r = rng.uniform(0, 1.02 * Nr * dzc, n)
    th = rng.uniform(0, 2 * np.pi, n)
    x, y = r * np.cos(th), r * np.sin(th)
    z = rng.uniform(0, Nz * dzc, n)
    ux, uy, uz = (rng.normal(size=n) for _ in range(3))
    ig = 1. / np.sqrt(1. + ux**2 + uy**2 + uz**2)
    w = rng.normal(size=n)
    dt = 0.5 * dzc / c
    geom = (1. / dzc, 0., Nz, 1. / dzc, 0., Nr)
    xr, yr, zr = x.copy(), y.copy(), z.copy()
    oracle.push_x(xr, yr, zr, ux, uy, uz, ig, dt, 1., 1., 1.)
    ref = oracle.cell_index(xr, yr, zr, *geom)
    t = hip.torch()
    ncell = Nz * (Nr + 1)
    src = [dev(hip, a) for a in (x, y, z, ux, uy, uz, w, ig)]
    dst = [t.empty_like(src[0]) for _ in range(8)]
    ci = t.empty(n, dtype=t.int32, device='cuda')
    si = t.empty(n, dtype=t.int32, device='cuda')
    pre = t.empty(ncell, dtype=t.int32, device='cuda')
    nb = int(hip.lib().fb_bin_sort_workspace_bytes(n, ncell))
    ws = t.empty(nb, dtype=t.uint8, device='cuda')
    p = hip.ptr
    if preranked:
        Nm = 2
        J = t.zeros((Nz, 3 * Nm, Nr), dtype=t.complex128, device='cuda')
        J2 = t.zeros_like(J)
        ruy = dev(hip, rng.uniform(-0.05, 0.05, Nr + 1))
        views = [J[:, k, :] for k in range(3 * Nm)]
        views2 = [J2[:, k, :] for k in range(3 * Nm)]
        q = -1.6e-19
        for shape in (1, 3):
            J.zero_(); J2.zero_()
            hip.check(hip.lib().fb_deposit_J_rank_next(
                shape, Nm, n, p(src[0]), p(src[1]), p(src[2]), p(src[6]), q, p(src[3]), p(src[4]),
                p(src[5]), p(src[7]), c, *geom, hip.ptr_array(views), 3 * Nm * Nr, 1, p(ruy), p(ruy),
                None, dt, 1., 1., 1., ncell, p(ws), nb, 0, hip.stream()), 'deposit_J_rank_next')
            hip.check(hip.lib().fb_deposit_J(
                shape, Nm, n, p(src[0]), p(src[1]), p(src[2]), p(src[6]), q, p(src[3]), p(src[4]),
                p(src[5]), p(src[7]), c, *geom, hip.ptr_array(views2), 3 * Nm * Nr, 1, None, p(ruy),
                p(ruy), None, hip.stream()), 'deposit_J')
            assert rel_err(host(J), host(J2)) < 1e-13
    hip.check(hip.lib().fb_push_x_bin_sort_particles(
        n, ncell, p(src[0]), p(src[1]), p(src[2]), p(src[3]), p(src[4]), p(src[5]), p(src[7]),
        c, dt, 1., 1., 1., *geom, 8, hip.ptr_array(src), hip.ptr_array(dst), p(ci), p(si), p(pre),
        p(ws), nb, preranked, hip.stream()), 'push_x_bin_sort')
    cis, sidx, prefix = host(ci), host(si), host(pre)
    assert np.all(np.diff(cis) >= 0)
    assert np.array_equal(np.sort(sidx), np.arange(n, dtype=np.int32))
    assert np.array_equal(cis, ref[sidx])
    assert np.array_equal(prefix, np.cumsum(np.bincount(ref, minlength=ncell)).astype(np.int32))
    for a, b in zip((xr, yr, zr, ux, uy, uz, w, ig), dst):
        assert np.array_equal(host(b), a[sidx])
    # the scatter pass leaves the per-cell counters of the workspace zeroed ...
    assert np.all(host(ws[:4 * ncell].view(t.int32)) == 0)
    if preranked:
        # ... so that the next rank pass may skip its memset (counts_are_zero = 1)
        hip.check(hip.lib().fb_deposit_J_rank_next(
            1, Nm, n, p(src[0]), p(src[1]), p(src[2]), p(src[6]), q, p(src[3]), p(src[4]),
            p(src[5]), p(src[7]), c, *geom, hip.ptr_array(views), 3 * Nm * Nr, 1, p(ruy), p(ruy),
            None, dt, 1., 1., 1., ncell, p(ws), nb, 1, hip.stream()), 'deposit_J_rank_next')
        dst2 = [t.empty_like(b) for b in dst]
        hip.check(hip.lib().fb_push_x_bin_sort_particles(
            n, ncell, p(src[0]), p(src[1]), p(src[2]), p(src[3]), p(src[4]), p(src[5]), p(src[7]),
            c, dt, 1., 1., 1., *geom, 8, hip.ptr_array(src), hip.ptr_array(dst2), p(ci), p(si), p(pre),
            p(ws), nb, 1, hip.stream()), 'push_x_bin_sort')
        assert np.array_equal(host(pre), prefix) and np.array_equal(host(ci), cis)
        sidx2 = host(si)
        for a, b in zip((xr, yr, zr, ux, uy, uz, w, ig), dst2):
            assert np.array_equal(host(b), a[sidx2])
    # the inputs are untouched
    for a, b in zip((x, y, z), src[:3]):
        assert np.array_equal(host(b), a)


@pytest.mark.parametrize('shape,Nm,preranked,nattr,records', [(1, 2, 1, 8, True), (1, 2, 0, 8, False),
                                                            (3, 2, 1, 14, False), (3, 4, 0, 8, False),
                                                            (1, 5, 1, 8, False), (1, 1, 1, 8, True)])
def test_push_sort_deposit_rho_fused(hip, oracle, shape, Nm, preranked, nattr, records):
    """fb_push_x_sort_deposit_rho == fb_push_x_bin_sort_particles followed by fb_deposit_rho:
    pushed positions bit-identical to the oracle push, same cells / prefix sums / permutation
    property, charge density equal to the two-call sequence AND to the oracle deposition of the
    pushed particles (1e-13, summation order).  Variants: preranked by fb_deposit_J_rank_next,
    extra attributes riding along (keep_fields_sorted), node-major record target, Nm > 4 (second
    deposition launch reads the arrays the first one wrote)."""
    from scipy.constants import c
    rng = np.random.default_rng(41 + Nm)
    n, Nz, Nr = 150001, 48, 24
    dzc = 0.2e-6
    # cell-sorted stream with thermal motion, as inside the PIC cycle
    r = rng.uniform(0, 1.02 * Nr * dzc, n)
    th = rng.uniform(0, 2 * np.pi, n)
    x, y = r * np.cos(th), r * np.sin(th)
    z = rng.uniform(0., Nz * dzc, n)       # pushed: within half a cell of the box (guard range)
    geom = (1. / dzc, 0., Nz, 1. / dzc, 0., Nr)
    o = np.argsort(oracle.cell_index(x, y, z, *geom), kind='stable')
    x, y, z = x[o], y[o], z[o]
    ux, uy, uz = (rng.normal(size=n) * 0.3 for _ in range(3))
    ig = 1. / np.sqrt(1. + ux**2 + uy**2 + uz**2)
    w = rng.uniform(0.5, 1.5, n)
    extra = [rng.normal(size=n) for _ in range(nattr - 8)]
    dt = 0.5 * dzc / c
    q = -1.6e-19
    xr, yr, zr = x.copy(), y.copy(), z.copy()
    oracle.push_x(xr, yr, zr, ux, uy, uz, ig, dt, 1., 1., 1.)
    ref_cell = oracle.cell_index(xr, yr, zr, *geom)
    t = hip.torch()
    p = hip.ptr
    ncell = Nz * (Nr + 1)
    host_attrs = [x, y, z, ux, uy, uz, w, ig] + extra
    src = [dev(hip, a) for a in host_attrs]
    ruy0 = dev(hip, rng.uniform(-0.05, 0.05, Nr + 1))
    ruyh = dev(hip, rng.uniform(-0.05, 0.05, Nr + 1))
    nb = int(hip.lib().fb_bin_sort_workspace_bytes(n, ncell))

    def target():
        if records:      # node-major records: (Nz, Nr, 4 Nm), rho of mode m at slot 4m+3
            rec = t.zeros((Nz, Nr, 4 * Nm), dtype=t.complex128, device='cuda')
            return rec, [rec[:, :, 4 * m + 3] for m in range(Nm)]
        g = t.zeros((Nz, Nm, Nr), dtype=t.complex128, device='cuda')
        return g, [g[:, m, :] for m in range(Nm)]

    def prerank(ws):
        J = t.zeros((Nz, 3 * Nm, Nr), dtype=t.complex128, device='cuda')
        hip.check(hip.lib().fb_deposit_J_rank_next(
            shape, Nm, n, p(src[0]), p(src[1]), p(src[2]), p(src[6]), q, p(src[3]), p(src[4]),
            p(src[5]), p(src[7]), c, *geom, hip.ptr_array([J[:, k, :] for k in range(3 * Nm)]),
            3 * Nm * Nr, 1, p(ruy0), p(ruyh), None, dt, 1., 1., 1., ncell, p(ws), nb, 0,
            hip.stream()), 'deposit_J_rank_next')
    # ---- fused
    ws = t.empty(nb, dtype=t.uint8, device='cuda')
    if preranked:
        prerank(ws)
    dst = [t.empty_like(a) for a in src]
    ci = t.empty(n, dtype=t.int32, device='cuda')
    si = t.empty(n, dtype=t.int32, device='cuda')
    pre = t.empty(ncell, dtype=t.int32, device='cuda')
    base, views = target()
    hip.check(hip.lib().fb_push_x_sort_deposit_rho(
        n, ncell, p(src[0]), p(src[1]), p(src[2]), p(src[3]), p(src[4]), p(src[5]), p(src[7]),
        c, dt, 1., 1., 1., *geom, nattr, hip.ptr_array(src), hip.ptr_array(dst), p(ci), p(si), p(pre),
        p(ws), nb, preranked, shape, Nm, q, hip.ptr_array(views), views[0].stride(0),
        views[0].stride(1), p(ruy0), p(ruyh), hip.stream()), 'push_x_sort_deposit_rho')
    cis, sidx, prefix = host(ci), host(si), host(pre)
    assert np.all(np.diff(cis) >= 0)
    assert np.array_equal(np.sort(sidx), np.arange(n, dtype=np.int32))
    assert np.array_equal(cis, ref_cell[sidx])
    assert np.array_equal(prefix, np.cumsum(np.bincount(ref_cell, minlength=ncell)).astype(np.int32))
    for a, b in zip([xr, yr, zr, ux, uy, uz, w, ig] + extra, dst):
        assert np.array_equal(host(b), a[sidx])
    assert np.all(host(ws[:4 * ncell].view(t.int32)) == 0)       # counters left zeroed
    for a, b in zip(host_attrs, src):
        assert np.array_equal(host(b), a)                        # inputs untouched
    # ---- the two-call sequence
    ws2 = t.empty(nb, dtype=t.uint8, device='cuda')
    if preranked:
        prerank(ws2)
    dst2 = [t.empty_like(a) for a in src]
    pre2 = t.empty(ncell, dtype=t.int32, device='cuda')
    hip.check(hip.lib().fb_push_x_bin_sort_particles(
        n, ncell, p(src[0]), p(src[1]), p(src[2]), p(src[3]), p(src[4]), p(src[5]), p(src[7]),
        c, dt, 1., 1., 1., *geom, nattr, hip.ptr_array(src), hip.ptr_array(dst2), None, None, p(pre2),
        p(ws2), nb, preranked, hip.stream()), 'push_x_bin_sort')
    base2, views2 = target()
    hip.check(hip.lib().fb_deposit_rho(shape, Nm, n, p(dst2[0]), p(dst2[1]), p(dst2[2]), p(dst2[6]), q,
                                       *geom, hip.ptr_array(views2), views2[0].stride(0),
                                       views2[0].stride(1), p(pre2), p(ruy0), p(ruyh), None,
                                       hip.stream()), 'deposit_rho')
    assert np.array_equal(host(pre2), prefix)
    assert rel_err(host(base), host(base2)) < 1e-13
    # ---- the oracle deposition of the pushed particles
    if Nm <= 4:
        glob = np.zeros((1, Nm, Nz + 4, Nr + 4), dtype=np.complex128)
        oracle.deposit_rho_global('linear' if shape == 1 else 'cubic', Nm, xr, yr, zr, w, q, *geom,
                                  host(ruy0), host(ruyh), 1, glob)
        for m in range(Nm):
            red = np.zeros((Nz, Nr), dtype=np.complex128)
            oracle.sum_reduce(glob, m, red)
            assert rel_err(host(views[m]), red) < 1e-13, m


@pytest.mark.parametrize('shape,Nm,records,engine,uscale',
                         [(1, 2, True, 0, 0.3), (1, 1, False, 0, 0.3), (3, 2, False, 0, 0.3),
                          (3, 4, False, 0, 0.3), (1, 3, True, 0, 0.3),
                          # engine 1: the two depositions one after the other
                          (1, 2, True, 1, 0.3), (1, 4, False, 0, 0.3),
                          # slow particles: < FB_PERM_SPLIT_AT (6) J-strays per chunk - the single
                          # traversal of k_perm_deposit_J_rho_merged with strays written one by one
                          (1, 2, True, 0, 0.01), (1, 3, True, 0, 0.01), (1, 1, False, 0, 0.01),
                          # fast ones: nearly every particle changes cell within the half push
                          (1, 2, True, 0, 3.0)])
def test_push_sort_deposit_J_rho_fused(hip, oracle, shape, Nm, records, engine, uscale):
    """fb_push_x_sort_deposit_J_rho == fb_deposit_J (positions before the push, its own zmin) then
    fb_push_x_sort_deposit_rho: same sorted particle arrays (bit-identical pushed positions), J
    and rho equal to the separate launches and to the oracle depositions (1e-13).
    `uscale` selects the branch of the merged linear engine (deposit.hip, k_perm_deposit_J_rho_merged):
    with u ~ 0.3 and dt = dz / 2c a quarter of the particles change their stencil within the push, ~16
    of 64 per chunk - the twice-traversed path (CycleDep::reduce_split, taken above FB_PERM_SPLIT_AT = 6
    J-strays per chunk); with u ~ 0.01 ~0.5 per chunk - the single traversal."""
    from scipy.constants import c
    rng = np.random.default_rng(51 + Nm)
    n, Nz, Nr = 120001, 40, 24
    dzc = 0.2e-6
    r = rng.uniform(0, 1.02 * Nr * dzc, n)
    th = rng.uniform(0, 2 * np.pi, n)
    x, y = r * np.cos(th), r * np.sin(th)
    z = rng.uniform(0., Nz * dzc, n)
    geom = (1. / dzc, 0., Nz, 1. / dzc, 0., Nr)
    o = np.argsort(oracle.cell_index(x, y, z, *geom), kind='stable')
    x, y, z = x[o], y[o], z[o]
    ux, uy, uz = (rng.normal(size=n) * uscale for _ in range(3))
    ig = 1. / np.sqrt(1. + ux**2 + uy**2 + uz**2)
    w = rng.uniform(0.5, 1.5, n)
    dt = 0.5 * dzc / c
    q = -1.6e-19
    xr, yr, zr = x.copy(), y.copy(), z.copy()
    oracle.push_x(xr, yr, zr, ux, uy, uz, ig, dt, 1., 1., 1.)
    ref_cell = oracle.cell_index(xr, yr, zr, *geom)
    t = hip.torch()
    p = hip.ptr
    ncell = Nz * (Nr + 1)
    host_attrs = [x, y, z, ux, uy, uz, w, ig]
    src = [dev(hip, a) for a in host_attrs]
    ruy0 = dev(hip, rng.uniform(-0.05, 0.05, Nr + 1))
    ruyh = dev(hip, rng.uniform(-0.05, 0.05, Nr + 1))
    nb = int(hip.lib().fb_bin_sort_workspace_bytes(n, ncell))

    def target():
        if records:      # node-major records: (Nz, Nr, 4 Nm): Jr, Jt, Jz, rho of mode m at 4m..4m+3
            rec = t.zeros((Nz, Nr, 4 * Nm), dtype=t.complex128, device='cuda')
            return rec, [rec[:, :, 4 * m + k] for m in range(Nm) for k in range(3)], \
                [rec[:, :, 4 * m + 3] for m in range(Nm)]
        g = t.zeros((Nz, 4 * Nm, Nr), dtype=t.complex128, device='cuda')
        return g, [g[:, 4 * m + k, :] for m in range(Nm) for k in range(3)], \
            [g[:, 4 * m + 3, :] for m in range(Nm)]
    # ---- fused
    ws = t.empty(nb, dtype=t.uint8, device='cuda')
    dst = [t.empty_like(a) for a in src]
    si = t.empty(n, dtype=t.int32, device='cuda')
    pre = t.empty(ncell, dtype=t.int32, device='cuda')
    base, jv, rv = target()
    hip.check(hip.lib().fb_push_x_sort_deposit_J_rho(
        n, ncell, p(src[0]), p(src[1]), p(src[2]), p(src[3]), p(src[4]), p(src[5]), p(src[7]),
        c, dt, 1., 1., 1., *geom, 8, hip.ptr_array(src), hip.ptr_array(dst), None, p(si), p(pre),
        p(ws), nb, 0, shape, Nm, q, 0., hip.ptr_array(jv), jv[0].stride(0), jv[0].stride(1),
        hip.ptr_array(rv), rv[0].stride(0), rv[0].stride(1), p(ruy0), p(ruyh), engine, hip.stream()),
        'push_x_sort_deposit_J_rho')
    sidx, prefix = host(si), host(pre)
    assert np.array_equal(np.sort(sidx), np.arange(n, dtype=np.int32))
    assert np.all(np.diff(ref_cell[sidx]) >= 0)
    assert np.array_equal(prefix, np.cumsum(np.bincount(ref_cell, minlength=ncell)).astype(np.int32))
    for a, b in zip([xr, yr, zr, ux, uy, uz, w, ig], dst):
        assert np.array_equal(host(b), a[sidx])
    # ---- separate launches
    base2, jv2, rv2 = target()
    hip.check(hip.lib().fb_deposit_J(
        shape, Nm, n, p(src[0]), p(src[1]), p(src[2]), p(src[6]), q, p(src[3]), p(src[4]),
        p(src[5]), p(src[7]), c, *geom, hip.ptr_array(jv2), jv2[0].stride(0), jv2[0].stride(1), None,
        p(ruy0), p(ruyh), None, hip.stream()), 'deposit_J')
    ws2 = t.empty(nb, dtype=t.uint8, device='cuda')
    dst2 = [t.empty_like(a) for a in src]
    hip.check(hip.lib().fb_push_x_sort_deposit_rho(
        n, ncell, p(src[0]), p(src[1]), p(src[2]), p(src[3]), p(src[4]), p(src[5]), p(src[7]),
        c, dt, 1., 1., 1., *geom, 8, hip.ptr_array(src), hip.ptr_array(dst2), None, p(si), p(pre),
        p(ws2), nb, 0, shape, Nm, q, hip.ptr_array(rv2), rv2[0].stride(0), rv2[0].stride(1),
        p(ruy0), p(ruyh), hip.stream()), 'push_x_sort_deposit_rho')
    a1, a2 = host(base), host(base2)
    for m in range(Nm):
        for k, nm in enumerate(('Jr', 'Jt', 'Jz')):
            g1, g2 = host(jv[3 * m + k]), host(jv2[3 * m + k])
            sc = max(np.abs(host(jv2[3 * mm + kk])).max() for mm in range(Nm) for kk in range(3))
            assert np.abs(g1 - g2).max() <= 1e-13 * sc, (m, nm)
        g1, g2 = host(rv[m]), host(rv2[m])
        sc = max(np.abs(host(rv2[mm])).max() for mm in range(Nm))
        assert np.abs(g1 - g2).max() <= 1e-13 * sc, (m, 'rho')
    # ---- oracle: J from the unpushed, rho from the pushed particles
    sh = 'linear' if shape == 1 else 'cubic'
    gl = oracle.deposit_J_global(sh, Nm, x, y, z, w, q, ux, uy, uz, ig, *geom, host(ruy0), host(ruyh), 1)
    gr = np.zeros((1, Nm, Nz + 4, Nr + 4), dtype=np.complex128)
    oracle.deposit_rho_global(sh, Nm, xr, yr, zr, w, q, *geom, host(ruy0), host(ruyh), 1, gr)
    for m in range(Nm):
        for k in range(3):
            red = np.zeros((Nz, Nr), dtype=np.complex128)
            oracle.sum_reduce(gl[k], m, red)
            sc = max(np.abs(gl[kk]).max() for kk in range(3))
            assert np.abs(host(jv[3 * m + k]) - red).max() <= 1e-13 * sc, (m, k)
        red = np.zeros((Nz, Nr), dtype=np.complex128)
        oracle.sum_reduce(gr, m, red)
        assert rel_err(host(rv[m]), red) < 1e-13 * max(1., np.abs(gr).max() / max(np.abs(red).max(), 1e-300)), m


def test_exchange_rccl_loopback(hip):
    """fb_comm_unique_id / fb_comm_init / fb_exchange (RCCL send/recv inside the library) on a
    1-rank communicator whose two neighbours are the rank itself - the 2-rank periodic ring
    collapsed onto one GPU: what is sent to the left arrives from the right and vice versa,
    stream-ordered after the kernel that produced the payload, without host synchronisation;
    then an open boundary on one side (no message posted there)."""
    import ctypes
    t = hip.torch()
    lib = hip.lib()
    idbuf = ctypes.create_string_buffer(128)
    hip.check(lib.fb_comm_unique_id(idbuf), 'fb_comm_unique_id')
    comm = ctypes.c_void_p()
    hip.check(lib.fb_comm_init(idbuf, 0, 1, ctypes.byref(comm)), 'fb_comm_init')
    n = 6 * 2 * 64 * 128                       # E,B guard block of C2: 64 rows x 12 fields x 128
    send_l = t.zeros(n, dtype=t.complex128, device='cuda')
    send_r = t.zeros(n, dtype=t.complex128, device='cuda')
    recv_l = t.full((n,), -1., dtype=t.complex128, device='cuda')
    recv_r = t.full((n,), -1., dtype=t.complex128, device='cuda')
    for it in range(3):
        send_l.copy_(t.arange(n, device='cuda') + 1000. * it)          # enqueued, not synchronised
        send_r.copy_(-(t.arange(n, device='cuda') + 1000. * it) * 1j)
        nb = n * 16
        hip.check(lib.fb_exchange(comm, 0, 0, hip.ptr(send_l), nb, hip.ptr(send_r), nb,
                                  hip.ptr(recv_l), nb, hip.ptr(recv_r), nb, hip.stream()), 'fb_exchange')
        # same-peer ring: my send-to-left is the peer's message from its right
        assert t.equal(recv_r, send_l) and t.equal(recv_l, send_r)
    # open boundary on the left: only the right pair is posted (to itself: send_right -> recv_right)
    recv_l.fill_(-1.)
    hip.check(lib.fb_exchange(comm, -1, 0, None, 0, hip.ptr(send_r), n * 16, None, 0,
                              hip.ptr(recv_r), n * 16, hip.stream()), 'fb_exchange')
    assert t.equal(recv_r, send_r) and bool((recv_l == -1.).all())
    assert lib.fb_exchange(None, 0, 0, None, 0, None, 0, None, 0, None, 0, hip.stream()) != 0
    hip.check(lib.fb_comm_destroy(comm), 'fb_comm_destroy')


def test_handover_pack_move_append(hip):
    """fb_handover_pack / _move / _append (the data movements of the particle hand-over, every
    attribute in one launch) against NumPy indexing."""
    rng = np.random.default_rng(21)
    t = hip.torch()
    n, nattr = 5000, 8
    a = rng.normal(size=(nattr, n))
    arrs = [dev(hip, np.concatenate([a[k], np.zeros(300)])) for k in range(nattr)]   # head-room
    idx = np.sort(rng.choice(n, 400, replace=False)).astype(np.int64)
    didx = dev(hip, idx)
    buf = t.zeros((nattr, 450), dtype=t.float64, device='cuda')
    hip.check(hip.lib().fb_handover_pack(idx.size, hip.ptr(didx), nattr, hip.ptr_array(arrs),
                                         hip.ptr(buf), buf.stride(0), hip.stream()), 'pack')
    assert np.array_equal(host(buf[:, :400]), a[:, idx]) and np.all(host(buf[:, 400:]) == 0)
    # holes among the first m slots are filled with the survivors of the tail
    m = n - idx.size
    tail_free = np.ones(n - m, bool)
    tail_free[idx[idx >= m] - m] = False
    src = (m + np.nonzero(tail_free)[0]).astype(np.int64)
    dst = idx[idx < m]
    assert src.size == dst.size
    dsrc, ddst = dev(hip, src), dev(hip, dst)          # (kept alive: the launch is asynchronous)
    hip.check(hip.lib().fb_handover_move(src.size, hip.ptr(dsrc), hip.ptr(ddst), nattr,
                                         hip.ptr_array(arrs), hip.stream()), 'move')
    expect = a.copy()
    expect[:, dst] = a[:, src]
    got = np.array([host(x)[:n] for x in arrs])
    assert np.array_equal(got[:, :m], expect[:, :m])
    assert sorted(got[0, :m].tolist()) == sorted(np.delete(a[0], idx).tolist())     # exactly the stayers
    arrivals = rng.normal(size=(nattr, 123))
    darr = dev(hip, arrivals)
    hip.check(hip.lib().fb_handover_append(123, m, nattr, hip.ptr_array(arrs), hip.ptr(darr),
                                           123, hip.stream()), 'append')
    got = np.array([host(x) for x in arrs])
    assert np.array_equal(got[:, m:m + 123], arrivals) and np.array_equal(got[:, :m], expect[:, :m])


def test_guard_buffers_and_damping(hip):
    """fb_guard_buffers (pack / replace / add of the guard rows of a field group, both z ends
    in one launch; boundaries/cuda_methods.py:12-484) and fb_damp_rows (:486-640) against
    NumPy slicing on a padded z-major slab."""
    t = hip.torch()
    p = hip.ptr
    rng = np.random.default_rng(77)
    Nz, NF, Nr, f0, nf, ng = 40, 7, 9, 2, 3, 4
    rs = NF * Nr + 8
    base = t.zeros(Nz * rs, dtype=t.complex128, device='cuda')
    slab = base.as_strided((Nz, NF, Nr), (rs, Nr, 1))
    h = rng.normal(size=(Nz, NF, Nr)) + 1j * rng.normal(size=(Nz, NF, Nr))
    slab.copy_(t.from_numpy(h))
    region = slab[:, f0:f0 + nf, :]
    ncontig = nf * Nr
    for nrows, zl, zr in ((ng, ng, Nz - 2 * ng), (2 * ng, 0, Nz - 2 * ng)):
        bl = t.empty((nrows, ncontig), dtype=t.complex128, device='cuda')
        br = t.empty_like(bl)
        hip.check(hip.lib().fb_guard_buffers(0, p(region), rs, ncontig, zl, zr, nrows, p(bl), p(br),
                                             hip.stream()), 'pack')
        assert np.array_equal(host(bl), h[zl:zl + nrows, f0:f0 + nf].reshape(nrows, ncontig))
        assert np.array_equal(host(br), h[zr:zr + nrows, f0:f0 + nf].reshape(nrows, ncontig))
        # one-sided (open end on the left): only the right buffer is written
        bl.zero_()
        hip.check(hip.lib().fb_guard_buffers(0, p(region), rs, ncontig, zl, zr, nrows, None, p(br),
                                             hip.stream()), 'pack right only')
        assert np.all(host(bl) == 0)
    # unpack: replace the outer ng rows, then add onto 2 ng rows
    inc_l = rng.normal(size=(2 * ng, ncontig)) + 1j * rng.normal(size=(2 * ng, ncontig))
    inc_r = rng.normal(size=(2 * ng, ncontig)) + 1j * rng.normal(size=(2 * ng, ncontig))
    dl, dr = dev(hip, inc_l), dev(hip, inc_r)
    hip.check(hip.lib().fb_guard_buffers(1, p(region), rs, ncontig, 0, Nz - ng, ng, p(dl), p(dr),
                                         hip.stream()), 'replace')
    exp = h.copy()
    exp[:ng, f0:f0 + nf] = inc_l[:ng].reshape(ng, nf, Nr)
    exp[Nz - ng:, f0:f0 + nf] = inc_r[:ng].reshape(ng, nf, Nr)
    assert np.array_equal(host(slab), exp)
    hip.check(hip.lib().fb_guard_buffers(2, p(region), rs, ncontig, 0, Nz - 2 * ng, 2 * ng, p(dl),
                                         p(dr), hip.stream()), 'add')
    exp[:2 * ng, f0:f0 + nf] += inc_l.reshape(2 * ng, nf, Nr)
    exp[Nz - 2 * ng:, f0:f0 + nf] += inc_r.reshape(2 * ng, nf, Nr)
    assert np.array_equal(host(slab), exp)
    assert np.all(host(base.as_strided((Nz, 8), (rs, 1), NF * Nr)) == 0)       # row padding untouched
    # damping of both ends / of one end
    damp_l, damp_r = rng.uniform(size=6), rng.uniform(size=5)
    d_damp_l, d_damp_r = dev(hip, damp_l), dev(hip, damp_r)
    hip.check(hip.lib().fb_damp_rows(p(region), rs, ncontig, p(d_damp_l), 6, p(d_damp_r), 5, Nz,
                                     hip.stream()), 'damp')
    exp[:6, f0:f0 + nf] *= damp_l[:, None, None]
    exp[Nz - 5:, f0:f0 + nf] *= damp_r[:, None, None]
    assert np.array_equal(host(slab), exp)
    hip.check(hip.lib().fb_damp_rows(p(region), rs, ncontig, None, 0, p(d_damp_r), 5, Nz,
                                     hip.stream()), 'damp right')
    exp[Nz - 5:, f0:f0 + nf] *= damp_r[:, None, None]
    assert np.array_equal(host(slab), exp)


@pytest.mark.parametrize('Nz,Nr', [(40, 24), (64, 128), (33, 70)])
def test_hankel_pm_to_rt(hip, Nz, Nr):
    """fb_hankel_pm_to_rt: r = p.Mp + m.Mm, t = i (p.Mp - m.Mm), z = z.M0 in one launch, on
    slab views (inverse transform of a vector field taken from z-real space)."""
    t = hip.torch()
    rng = np.random.default_rng(12)
    nf = 6
    src = rng.normal(size=(Nz, nf, Nr)) + 1j * rng.normal(size=(Nz, nf, Nr))
    mats = [rng.normal(size=(Nr, Nr)) for _ in range(nf)]
    d_src = dev(hip, src)
    d_out = t.zeros((Nz, nf + 1, Nr), dtype=t.complex128, device='cuda')
    d_mats = [dev(hip, M) for M in mats]
    outs = [d_out[:, j, :] for j in range(nf)]
    ins, in2, o1, o2, m1, m2 = [], [], [], [], [], []
    for g in range(0, nf, 3):
        ins += [d_src[:, g, :], d_src[:, g + 2, :]]
        in2 += [d_src[:, g + 1, :], None]
        o1 += [outs[g], outs[g + 2]]
        o2 += [outs[g + 1], None]
        m1 += [d_mats[g], d_mats[g + 2]]
        m2 += [d_mats[g + 1], None]
    hip.check(hip.lib().fb_hankel_pm_to_rt(
        len(ins), hip.ptr_array(ins), hip.ptr_array(in2), nf * Nr, hip.ptr_array(o1),
        hip.ptr_array(o2), (nf + 1) * Nr, hip.ptr_array(m1), hip.ptr_array(m2), 1.0, Nz, Nr,
        hip.stream()), 'fb_hankel_pm_to_rt')
    got = host(d_out)
    for g in range(nf // 3):
        p_, m_, z_ = (src[:, 3 * g + k, :] @ mats[3 * g + k] for k in range(3))
        assert rel_err(got[:, 3 * g, :], p_ + m_) < TOL
        assert rel_err(got[:, 3 * g + 1, :], 1j * (p_ - m_)) < TOL
        assert rel_err(got[:, 3 * g + 2, :], z_) < TOL
    assert np.all(got[:, nf, :] == 0)
