"""Host-side logic of the one-pass particle cycle (no GPU): re-keying of the home cells by the moving
window (Particles._home_shift), the sort policy fed by the pass's counters at a fixed lag (stray share,
bad chunks, suspension after a bad first pass), the mode in which the forward Hankel transform of J,
rho_next is deferred (Simulation._hankel_deferral), and tools/make_clamp.py.  The kernels themselves:
tests/test_gpu_onepass.py, test_gpu_lwfa.py, test_gpu_configs.py."""
import json
import os
import subprocess
import sys
import types
import numpy as np
import torch
import helpers
from conftest import ROOT


def _small_sim():
    return helpers.uniform_plasma_sim(8, 4, 2, (1, 1, 4), 'linear', seed=1)


class _Ev(object):
    """Stands in for the CUDA event of a counter read-back."""
    def __init__(self):
        self.waited = 0

    def synchronize(self):
        self.waited += 1


def test_home_cells_are_rekeyed_by_whole_cells_of_window_motion():
    sim = _small_sim()
    sp, g0 = sim.ptcl[0], sim.fld.interp[0]
    geom = lambda zmin: (zmin, g0.invdz, g0.Nz, g0.rmin, g0.invdr, g0.Nr)
    sp._home_geom = None
    assert sp._home_shift(g0) is None                          # no sort has recorded home cells
    sp._home_geom = geom(g0.zmin)
    assert sp._home_shift(g0) == 0
    # the window has advanced 3 cells since the sort: a particle that stayed is 3 rows further down
    sp._home_geom = geom(g0.zmin - 3 * g0.dz)
    assert sp._home_shift(g0) == 3 * (g0.Nr + 1)
    sp._home_geom = geom(g0.zmin + 2 * g0.dz * (1 + 1e-12))   # rounding of the accumulated zmin
    assert sp._home_shift(g0) == -2 * (g0.Nr + 1)
    # not a whole number of cells, another grid, further than the grid is long: re-sort
    sp._home_geom = geom(g0.zmin - 0.4 * g0.dz)
    assert sp._home_shift(g0) is None
    sp._home_geom = (g0.zmin, g0.invdz, g0.Nz + 1, g0.rmin, g0.invdr, g0.Nr)
    assert sp._home_shift(g0) is None
    sp._home_geom = geom(g0.zmin - (g0.Nz + 1) * g0.dz)
    assert sp._home_shift(g0) is None
    # ... and that is what makes the next iteration a sorting one
    sp._home_valid = True
    sp._cycle_since_sort = 0
    assert sp._cycle_needs_sort(g0) is True
    sp._home_geom = geom(g0.zmin - g0.dz)
    assert sp._cycle_needs_sort(g0) is False


def _with_stats(sp):
    sp._cycle_stats = [None, [torch.zeros(1024, dtype=torch.int64) for _ in range(3)], [], (0, 0), 0]
    sp._home_valid = True
    sp._home_geom = None
    return sp._cycle_stats


def _measured(sp, strays, bad_chunks, r, ntot=6400):
    """What Particles.cycle leaves behind a pass: cumulative counters in the next host buffer."""
    st = sp._cycle_stats
    buf = st[1][st[4] % 3]
    prev = st[1][(st[4] - 1) % 3] if st[4] else torch.zeros(1024, dtype=torch.int64)
    buf.copy_(prev)
    buf[0] += strays
    buf[512] += bad_chunks
    st[4] += 1
    ev = _Ev()
    st[2].append((ev, ntot, r, buf, sp.cycle_sorts))
    return ev


def test_sort_policy_reads_the_counters_at_a_fixed_lag():
    sim = _small_sim()
    sp = sim.ptcl[0]
    _with_stats(sp)
    sp.cycle_stray_limit, sp.cycle_bad_limit = 0.12, 0.01
    e1 = _measured(sp, 6400 // 4, 0, r=1)             # 25 % strays
    sp._cycle_poll()                                  # lag 1: the newest read-back stays in flight
    assert e1.waited == 0 and sp.cycle_stray_fraction is None
    e2 = _measured(sp, 64, 0, r=2)                    # 1 %
    sp._cycle_poll()
    assert e1.waited == 1 and e2.waited == 0
    assert abs(sp.cycle_stray_fraction - 0.25) < 1e-12 and sp.cycle_bad_fraction == 0.
    sp._cycle_poll(lag=0)
    assert e2.waited == 1 and abs(sp.cycle_stray_fraction - 0.01) < 1e-12      # differences of the cumulative counters
    assert sp.cycle_last_stray_fraction == sp.cycle_stray_fraction
    # a pass of the order BEFORE the latest sort does not speak for the present order
    e3 = _measured(sp, 6400, 0, r=3)
    sp._after_home_sort()
    assert sp.cycle_stray_fraction is None
    sp._cycle_poll(lag=0)
    assert e3.waited == 1 and sp.cycle_stray_fraction is None and sp.cycle_last_stray_fraction == 1.0


def test_bad_chunks_ask_for_a_sort_and_a_bad_first_pass_suspends_the_one_pass_form():
    sim = _small_sim()
    sp, fld = sim.ptcl[0], sim.fld
    g0 = fld.interp[0]
    _with_stats(sp)
    sp._home_geom = (g0.zmin, g0.invdz, g0.Nz, g0.rmin, g0.invdr, g0.Nr)
    sp.cycle_stray_limit, sp.cycle_bad_limit, sp.cycle_suspend_iterations = 0.5, 0.01, 4
    sp.cycle_sort_period = 100
    sp._cycle_since_sort = 2
    # 3 of 100 chunks hold > 16 strays, seen at r = 2: a sort, no suspension
    _measured(sp, 60, 3, r=2)
    sp._cycle_poll(lag=0)
    assert sp.cycle_bad_fraction == 0.03 and sp._cycle_suspended == 0
    assert sp.cycle_wants_sort(fld) is True
    sp._after_home_sort()
    assert sp.cycle_wants_sort(fld) is False           # fresh order, nothing measured on it yet
    # the FIRST pass after the sort reports the same: the plasma does not fit the one-pass form now
    _measured(sp, 60, 3, r=1)
    _measured(sp, 0, 0, r=2)
    asked = []
    for _ in range(6):
        asked.append(sp.cycle_wants_sort(fld))
        if asked[-1]:
            sp._after_home_sort()                      # (the two-pass iteration that follows sorts)
    assert asked == [True] * 4 + [False, False]        # 4 suspended iterations, then one pass probes again
    assert sp._cycle_suspended == 0
    # a neutral or empty species never asks
    sp.q = 0
    assert sp.cycle_wants_sort(fld) is False


def test_forward_hankel_is_deferred_to_the_launch_that_fits_the_domain():
    sim = _small_sim()
    sim._in_step = True
    assert sim._hankel_deferral() is True              # single periodic domain: whole spectral cycle
    sim.comm = types.SimpleNamespace(size=2, nz_damp=0, moving_win=None)
    assert sim._hankel_deferral() == 'correct'         # decomposed: transform + correction
    sim.reference_sequence = True
    assert sim._hankel_deferral() is False
    sim.reference_sequence = False
    sim.fld.current_correction = 'cross-deposition'
    assert sim._hankel_deferral() is False
    sim.fld.current_correction = 'curl-free'
    sim._in_step = False
    assert sim._hankel_deferral() is False


def test_make_clamp_keeps_the_worst_figure_of_every_check(tmp_path):
    a = {'x s1': {'achieved': 1e-14, 'bound': 1e-12}, 'y': {'achieved': 3e-15, 'bound': 1e-13},
         'c4 rank-loss retries (count)': {'achieved': 0.0, 'bound': 3.5}}
    b = {'x s1': {'achieved': 4e-15, 'bound': 1e-12}, 'z': 2e-16}
    pa, pb, out = tmp_path / 'a.json', tmp_path / 'b.json', tmp_path / 'o.json'
    pa.write_text(json.dumps(a)); pb.write_text(json.dumps(b))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_clamp.py'), str(out), str(pa), str(pb)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.loads(out.read_text()) == {'x s1': 1e-14, 'y': 3e-15, 'z': 2e-16}
    # and conftest reads the newest committed file of that kind
    import conftest
    newest = conftest.newest_clamp_file()
    stored = {k: v for k, v in json.load(open(newest)).items() if not k.startswith('__')}
    assert conftest._MEASURED == stored and len(conftest._MEASURED) > 250
    # a new clamp never loosens an existing check silently: min(previous, new) unless a reason is given
    prev, why, out2 = tmp_path / 'prev.json', tmp_path / 'why.json', tmp_path / 'o2.json'
    prev.write_text(json.dumps({'x s1': 5e-15, 'y': 1e-14, 'gone': 7e-15}))
    why.write_text(json.dumps({'y': 'not this one'}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_clamp.py'), str(out2), '--previous', str(prev),
                        '--looser', str(why), str(pa), str(pb)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.loads(out2.read_text()) == {'gone': 7e-15, 'x s1': 5e-15, 'y': 3e-15, 'z': 2e-16}
    why.write_text(json.dumps({'x s1': 'summation order of the new kernel'}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_clamp.py'), str(out2), '--previous', str(prev),
                        '--looser', str(why), str(pa), str(pb)], capture_output=True, text=True)
    got = json.loads(out2.read_text())
    assert got['x s1'] == 1e-14 and got['__looser__']['x s1']['previous'] == 5e-15


def test_clamp_file_is_chosen_by_round_number(tmp_path, monkeypatch):
    import conftest
    for name in ('achieved_r99.json', 'achieved_r100.json', 'achieved_r05.json'):
        (tmp_path / name).write_text('{}')
    monkeypatch.setattr(conftest, 'GOLDEN', str(tmp_path))
    assert os.path.basename(conftest.newest_clamp_file()) == 'achieved_r100.json'


def test_every_species_is_asked_every_iteration():
    """Simulation.step asked the species with any(generator) - once a species wanted a
    sort the later ones were neither polled nor counted down.  Now every species is asked."""
    sim = _small_sim()
    a = sim.ptcl[0]
    b = sim.add_new_species(q=a.q, m=a.m)
    for k in ('x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w'):
        setattr(b, k, getattr(a, k).copy())
    b.Ntot = a.Ntot
    calls = []
    a.cycle_wants_sort = lambda fld: calls.append('a') or True
    b.cycle_wants_sort = lambda fld: calls.append('b') or False
    src = open(os.path.join(ROOT, 'fbpic_amd', 'main.py')).read()
    assert 'any([sp.cycle_wants_sort(fld) for sp in ptcl])' in src
    assert any([sp.cycle_wants_sort(sim.fld) for sp in sim.ptcl]) and calls == ['a', 'b']
