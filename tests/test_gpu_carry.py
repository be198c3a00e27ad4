"""State carried from one Simulation.step call to the next (fbpic_amd/main.py, `_carry_signature`):
`for _ in range(n): sim.step(1)` must give what `sim.step(n)` gives - the reference's
"the user may have changed the particles" work at i_step == 0 (main.py:435-451) is only skipped
while nobody touched the tensors - and must not cost more (the timing half of this is
tests/test_gpu_configs.py::test_c2_step1_loop_costs_what_stepN_costs)."""
import numpy as np
import pytest
import helpers
from conftest import achieved

pytestmark = pytest.mark.gpu


def _state(sim):
    out = {}
    for m in range(sim.fld.Nm):
        for k in helpers.INTERP:
            out['%s%d' % (k, m)] = getattr(sim.fld.interp[m], k)
    out = {k: v.detach().cpu().numpy().copy() for k, v in out.items()}
    s = sim.ptcl[0]
    # (Ex..Bz keep the order of the gather that wrote them, not the order of the sort that
    # followed it - in the reference's GPU path too -, so only the eight state arrays compare)
    P = np.array([getattr(s, k).detach().cpu().numpy() for k in helpers.PTCL[:8]])
    o = np.lexsort((P[2], P[1], P[0], P[7]))
    out['ptcl'] = P[:, o]
    return out


def _compare(a, b, tag, tol):
    for grp in ('E', 'B', 'J', 'r'):
        keys = [k for k in a if k != 'ptcl' and k[0] == grp]
        scale = max(np.abs(b[k]).max() for k in keys)
        if scale > 0:
            achieved('%s fields %s' % (tag, grp), max(np.abs(a[k] - b[k]).max() for k in keys) / scale, tol)
    for j, k in enumerate(helpers.PTCL[:8]):
        sc = np.abs(b['ptcl'][j]).max()
        if sc > 0:
            achieved('%s particles' % tag, np.abs(a['ptcl'][j] - b['ptcl'][j]).max() / sc, tol)


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_step1_loop_equals_stepN(shape):
    from fbpic_amd.main import GpuMemoryManager
    n = 6
    res = {}
    for mode in ('one_call', 'loop_carried', 'loop_not_carried'):
        sim = helpers.uniform_plasma_sim(64, 32, 2, (2, 2, 4), shape, seed=3, u_th=0.05)
        sim.carry_state_between_calls = (mode != 'loop_not_carried')
        with GpuMemoryManager(sim):
            if mode == 'one_call':
                sim.step(n)
            else:
                flags = []
                for _ in range(n):
                    sim.step(1)
                    flags.append(sim._last_call_carried)
                assert flags == [False] + [mode == 'loop_carried'] * (n - 1), flags
            res[mode] = _state(sim)
    # carried calls run the interior-iteration sequence of the single call: same arithmetic
    # (differences = the order of the deposition atomics)
    _compare(res['loop_carried'], res['one_call'], 'carry: step(1) loop vs step(n)', 2e-12)
    # ... and the reference's per-call sequence (rho_prev deposited again, E, B transformed
    # again at the start of every call) agrees with it to rounding
    _compare(res['loop_not_carried'], res['one_call'], 'carry: reference per-call sequence vs step(n)', 2e-12)


def test_carry_dropped_when_the_user_touches_the_data():
    """An in-place change of a particle or field tensor between two calls (torch bumps the
    tensor's version counter) brings the reference's first-iteration work back: results equal
    those of a simulation that never carries anything."""
    from fbpic_amd.main import GpuMemoryManager
    res = {}
    for carry in (True, False):
        sim = helpers.uniform_plasma_sim(64, 32, 2, (2, 2, 4), 'linear', seed=4, u_th=0.05)
        sim.carry_state_between_calls = carry
        with GpuMemoryManager(sim):
            sim.step(2)
            sim.step(1)
            assert sim._last_call_carried is carry
            sim.ptcl[0].w *= 1.5                       # heavier macroparticles: rho_prev is stale
            sim.step(1)
            assert sim._last_call_carried is False
            sim.ptcl[0].q *= 1.                        # unchanged value: still carried
            sim.step(1)
            assert sim._last_call_carried is carry
            sim.ptcl[0].q *= 0.5                       # a plain attribute, no tensor: noticed too
            sim.step(1)
            assert sim._last_call_carried is False
            sim.step(1)
            assert sim._last_call_carried is carry
            sim.fld.interp[1].Er[3:9, 2:5] += 1.e9     # E changed on the grid: must be re-transformed
            sim.step(1)
            assert sim._last_call_carried is False
            sim.step(2)
            res[carry] = _state(sim)
    _compare(res[True], res[False], 'carry dropped on user modification', 2e-12)


def test_deferred_sources_are_what_the_eager_tail_gives():
    """J and rho on the interpolation grid after a call: computed on first read."""
    from fbpic_amd.main import GpuMemoryManager
    vals = {}
    for carry in (True, False):
        sim = helpers.uniform_plasma_sim(32, 16, 2, (2, 2, 4), 'linear', seed=5, u_th=0.05)
        sim.carry_state_between_calls = carry
        with GpuMemoryManager(sim):
            sim.step(3)
            assert (sim.fld._deferred_sources is not None) is carry
            vals[carry] = [sim.fld.interp[1].Jz.detach().cpu().numpy().copy(),
                           sim.fld.interp[0].rho.detach().cpu().numpy().copy()]
            assert sim.fld._deferred_sources is None
        # leaving the manager copies the grids to the host: NumPy arrays, as in the reference
        assert isinstance(sim.fld.interp[0].Jr, np.ndarray)
    for a, b in zip(vals[True], vals[False]):
        achieved('deferred J / rho vs eager', np.abs(a - b).max() / np.abs(b).max(), 2e-14)


@pytest.mark.parametrize('shape', ['linear', 'cubic'])
def test_deferred_particle_fields_vs_oracle(oracle, shape):
    """species.Ex ... Bz after a call whose last gather did not store them: evaluated on first
    read from the saved grids and the stepped-back positions (Particles.defer_fields).  The
    oracle keeps its particles in their initial order and stores E, B at every gather
    (reference semantics), so all 14 arrays can be compared particle by particle."""
    from fbpic_amd.main import GpuMemoryManager
    sim = helpers.uniform_plasma_sim(32, 16, 2, (2, 2, 4), shape, seed=6, u_th=0.05)
    orc = helpers.oracle_from_sim(oracle, sim)
    with GpuMemoryManager(sim):
        sim.step(2)
        sim.step(1)
        s = sim.ptcl[0]
        assert s._deferred_fields is not None and 'Ex' not in s.__dict__
        got = np.array([getattr(s, k).detach().cpu().numpy() for k in helpers.PTCL])
        assert s._deferred_fields is None
    orc.step(3)
    o = orc.species[0]
    ref = np.array([o[k] for k in helpers.PTCL])
    o1 = np.lexsort((ref[2], ref[1], ref[0], ref[7]))
    o2 = np.lexsort((got[2], got[1], got[0], got[7]))
    for j, k in enumerate(helpers.PTCL):
        grp = slice(8, 11) if 8 <= j < 11 else (slice(11, 14) if j >= 11 else slice(j, j + 1))
        sc = np.abs(ref[grp]).max()
        achieved(None, np.abs(got[j][o2] - ref[j][o1]).max() / sc, 1e-12,
                 'state' if j < 8 else 'E, B on the particles')          # measured 9.4e-14
    # leaving the manager with deferred fields pending: they are evaluated for the host copy
    sim2 = helpers.uniform_plasma_sim(32, 16, 2, (2, 2, 4), shape, seed=6, u_th=0.05)
    with GpuMemoryManager(sim2):
        sim2.step(3)
        assert sim2.ptcl[0]._deferred_fields is not None
        q0 = sim2.ptcl[0].q
        sim2.ptcl[0].q = 2 * q0          # a plain attribute, no tensor: still noticed
        sim2.ptcl[0].q = q0
    got2 = np.array([getattr(sim2.ptcl[0], k) for k in helpers.PTCL])
    o3 = np.lexsort((got2[2], got2[1], got2[0], got2[7]))
    for j in range(8, 14):
        grp = slice(8, 11) if j < 11 else slice(11, 14)
        achieved(None, np.abs(got2[j][o3] - ref[j][o1]).max() / np.abs(ref[grp]).max(), 1e-12,
                 'E, B on the particles (host copy)')
