"""The fused r-spectral part of a step (csrc/spectral_cycle.hip, fb_spect_cycle_standard;
Fields.spect_cycle): forward Hankel transform of J | rho_next + PSATD step + inverse Hankel
transform of E, B in one launch, against the three entry points it replaces (same MFMA sums in
the same order: agreement to rounding) and, through Simulation.step, against the separate
launches and the oracle."""
import numpy as np
import pytest
from scipy.constants import c, epsilon_0, mu_0
from conftest import achieved

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from fbpic_amd import _capi
    _capi.require_device()
    return _capi


def dev(hip, a):
    return hip.to_device(np.ascontiguousarray(a))


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('Nz,Nr,Nm,correct,utr,filt', [(64, 128, 2, 1, 0, True), (40, 50, 3, 1, 1, True),
                                                       (24, 128, 1, 0, 0, False), (19, 33, 2, 1, 0, True),
                                                       # correct = 2: forward transform + correction only
                                                       # (the launch of a decomposed domain)
                                                       (72, 128, 2, 2, 0, True), (19, 33, 3, 2, 0, False)])
def test_spect_cycle_equals_the_three_entry_points(hip, Nz, Nr, Nm, correct, utr, filt):
    rng = np.random.default_rng(100 * Nr + Nm)
    t = hip.torch()
    pa, st = hip.ptr_array, hip.stream()
    lib = hip.lib()
    dt = 6.67e-16

    def cplx(*shape):
        return rng.normal(size=shape) + 1j * rng.normal(size=shape)
    src_h = cplx(Nz, 4 * Nm, Nr)                       # [J m0 r,t,z | J m1 ... | rho m0 ...] in (kz, r)
    spect_h = cplx(Nz, 11 * Nm, Nr) * 1e3
    mats_f = [dev(hip, rng.normal(size=(Nr, Nr))) for _ in range(3 * Nm)]
    mats_i = [dev(hip, rng.normal(size=(Nr, Nr))) for _ in range(3 * Nm)]
    invvol = [dev(hip, rng.uniform(0.5, 2., Nr)) for _ in range(Nm)]
    fz = [dev(hip, rng.uniform(0., 1., Nz)) for _ in range(Nm)]
    fr = [dev(hip, rng.uniform(0., 1., Nr)) for _ in range(Nm)]
    tabs = []
    for m in range(Nm):
        kz = np.repeat(rng.normal(size=Nz)[:, None] * 1e6, Nr, 1)
        kr = np.repeat(np.abs(rng.normal(size=Nr))[None, :] * 1e6, Nz, 0)
        w = c * np.sqrt(kz**2 + kr**2)
        tabs.append([dev(hip, x) for x in (
            rng.normal(size=(Nz, Nr)), rng.normal(size=(Nz, Nr)), rng.normal(size=(Nz, Nr)) * 1e-3,
            np.cos(w * dt), np.sin(w * dt) / w, kr, kz, 1. / (kz**2 + kr**2))])
    tables = [x for tb in tabs for x in tb]

    def run(fused):
        src = dev(hip, src_h)
        spect = dev(hip, spect_h)
        out = t.zeros((Nz, 6 * Nm, Nr), dtype=t.complex128, device='cuda')
        sv = [src[:, j, :] for j in range(4 * Nm)]
        fields = [spect[:, 11 * m + i, :] for m in range(Nm) for i in range(11)]
        ov = [out[:, j, :] for j in range(6 * Nm)]
        if fused:
            srcs, outs = [], []
            for m in range(Nm):
                srcs += sv[3 * m:3 * m + 3] + [sv[3 * Nm + m]]
                outs += ov[3 * m:3 * m + 3] + ov[3 * Nm + 3 * m:3 * Nm + 3 * m + 3]
            hip.check(lib.fb_spect_cycle_standard(
                Nm, pa(srcs), src.stride(0), pa(invvol), pa(mats_f), pa(mats_i),
                pa(fz) if filt else None, pa(fr) if filt else None, pa(fields), spect.stride(0), pa(tables),
                dt, correct, utr, c, epsilon_0, mu_0, pa(outs), out.stride(0), Nz, Nr, st), 'spect_cycle')
        else:
            # forward: jobs in slab order, p / m pairs of every mode, then the scalars
            ins, in2, sgn, outf, mats, sk, ffz, ffr = [], [], [], [], [], [], [], []
            for m in range(Nm):
                r_, t_, z_ = sv[3 * m:3 * m + 3]
                ins += [r_, r_, z_]; in2 += [t_, t_, None]; sgn += [-1., 1., 0.]
                outf += [spect[:, 11 * m + 6 + k, :] for k in range(3)]
                mats += mats_f[3 * m:3 * m + 3]
                sk += [invvol[m]] * 3; ffz += [fz[m]] * 3; ffr += [fr[m]] * 3
            for m in range(Nm):
                ins.append(sv[3 * Nm + m]); in2.append(None); sgn.append(0.)
                outf.append(spect[:, 11 * m + 10, :]); mats.append(mats_f[3 * m + 2])
                sk.append(invvol[m]); ffz.append(fz[m]); ffr.append(fr[m])
            import ctypes
            sg = (ctypes.c_double * len(sgn))(*sgn)
            hip.check(lib.fb_hankel_rt_to_pm_scaled(
                4 * Nm, pa(ins), pa(in2), sg, src.stride(0), pa(outf), spect.stride(0), pa(mats), pa(sk),
                pa(ffz) if filt else None, pa(ffr) if filt else None, 1.0, Nz, Nr, st), 'hankel fwd')
            hip.check(lib.fb_psatd_step_standard(Nm, pa(fields), spect.stride(0), pa(tables), dt, correct,
                                                 utr, c, epsilon_0, mu_0, Nz, Nr, st), 'psatd')
            if correct == 2:
                return host(spect), host(out)
            inp, outs, mi = [], [], []
            for m in range(Nm):
                inp += [spect[:, 11 * m + i, :] for i in range(6)]
                outs += ov[3 * m:3 * m + 3] + ov[3 * Nm + 3 * m:3 * Nm + 3 * m + 3]
                mi += mats_i[3 * m:3 * m + 3] * 2
            hip.check(lib.fb_hankel(6 * Nm, pa(inp), spect.stride(0), pa(outs), out.stride(0), pa(mi), 1.0,
                                    Nz, Nr, st), 'hankel inv')
        return host(spect), host(out)
    s1, o1 = run(True)
    s0, o0 = run(False)
    if not correct:
        # without the correction fb_psatd_step_standard leaves J as the forward transform wrote it: same
        pass
    worst = 0.
    for m in range(Nm):
        for i in range(11):
            a, b = s1[:, 11 * m + i, :], s0[:, 11 * m + i, :]
            sc = max(np.abs(b).max(), 1e-300)
            worst = max(worst, np.abs(a - b).max() / sc)
    achieved(None, worst, 1e-13, 'spectral slab vs separate')
    if correct == 2:
        assert not o1.any() and not o0.any()          # nothing but J, rho_next was written
        assert np.array_equal(s1[:, [11 * m + i for m in range(Nm) for i in (0, 1, 2, 3, 4, 5, 9)], :],
                              spect_h[:, [11 * m + i for m in range(Nm) for i in (0, 1, 2, 3, 4, 5, 9)], :])
    else:
        achieved(None, np.abs(o1 - o0).max() / np.abs(o0).max(), 1e-13, 'E, B in (kz, r) vs separate')


@pytest.mark.parametrize('Nm', [1, 2, 3])
def test_step_with_fused_spectral_cycle_equals_separate_launches(hip, oracle, Nm):
    """Simulation.step with Fields.fuse_spectral_cycle on / off and against the oracle."""
    import helpers
    sims = []
    for fuse in (True, False):
        sim = helpers.uniform_plasma_sim(64, 32, Nm, (2, 2, 4), 'linear', seed=11, u_th=0.05)
        sim.fld.fuse_spectral_cycle = fuse
        if fuse:
            ref = helpers.oracle_from_sim(oracle, sim)
        sim.step(5)
        sims.append(sim)
    ref.step(5)
    a, b = sims
    e1 = e2 = 0.
    for m in range(Nm):
        for k in helpers.INTERP:
            grp = [kk for kk in helpers.INTERP if kk[0] == k[0]]
            sc = max(np.abs(ref.interp[mm][kk]).max() for mm in range(Nm) for kk in grp)
            e1 = max(e1, np.abs(getattr(a.fld.interp[m], k) - getattr(b.fld.interp[m], k)).max() / sc)
            e2 = max(e2, np.abs(getattr(a.fld.interp[m], k) - ref.interp[m][k]).max() / sc)
    achieved(None, e1, 1e-12, 'fields fused vs separate s5')
    achieved(None, e2, 5e-12, 'fields fused vs oracle s5')
