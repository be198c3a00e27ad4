"""Shared helpers for the parity tests: build the oracle's whole-cycle simulation from the
host-side tables of an fbpic_amd `Simulation` (same inputs, independent compute path)."""
import numpy as np

PTCL = ['x', 'y', 'z', 'ux', 'uy', 'uz', 'inv_gamma', 'w', 'Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz']
INTERP = ['Er', 'Et', 'Ez', 'Br', 'Bt', 'Bz', 'Jr', 'Jt', 'Jz', 'rho']
SPECT = ['Ep', 'Em', 'Ez', 'Bp', 'Bm', 'Bz', 'Jp', 'Jm', 'Jz', 'rho_prev', 'rho_next']


def oracle_from_sim(orc, sim, nthreads=1):
    """OracleSim with a private copy of the (host) state of `sim`."""
    return orc.from_sim(sim, nthreads=nthreads)


def set_species_state(species, arr14):
    for k, v in zip(PTCL, arr14):
        setattr(species, k, np.array(v, dtype=np.float64, copy=True))
    species.Ntot = arr14.shape[1]


def uniform_plasma_sim(Nz, Nr, Nm, ppc, shape, seed=0, dz=0.2e-6, n_e=2e24, u_th=0.01,
                       n_order=-1, n_guard=None):
    """Synthetic uniform-plasma input of SURVEY.md 8d (C2 family) at any size."""
    from scipy.constants import c
    from fbpic_amd.main import Simulation
    zmax, rmax = Nz * dz, Nr * dz
    np.random.seed(seed)
    sim = Simulation(Nz, zmax, Nr, rmax, Nm, dz / c, 0., zmax, 0., rmax, ppc[0], ppc[1], ppc[2],
                     n_e, particle_shape=shape, n_order=n_order, n_guard=n_guard)
    rng = np.random.default_rng(seed + 1)
    s = sim.ptcl[0]
    s.ux = rng.normal(0., u_th, s.Ntot)
    s.uy = rng.normal(0., u_th, s.Ntot)
    s.uz = rng.normal(0., u_th, s.Ntot)
    s.inv_gamma = 1. / np.sqrt(1 + s.ux**2 + s.uy**2 + s.uz**2)
    return sim


def build_from_golden(g, name):
    from fbpic_amd.main import Simulation
    Nz, Nr, Nm = int(g['Nz']), int(g['Nr']), int(g['Nm'])
    zmin = float(g['zmin']) if 'zmin' in g.files else 0.
    n_order = int(g['n_order']) if 'n_order' in g.files else -1
    shape = str(g['shape']) if 'shape' in g.files else ('linear' if 'linear' in name or 'lin' in name else 'cubic')
    extra = {}
    if 'v_comoving' in g.files and np.isfinite(float(g['v_comoving'])):
        # Galilean / comoving-current PSATD
        extra = dict(v_comoving=float(g['v_comoving']), use_galilean=bool(g['use_galilean']))
    if 'cross' in name:                  # cross-deposition current correction
        extra['current_correction'] = 'cross-deposition'
    sim = Simulation(Nz, float(g['zmax']), Nr, float(g['rmax']), Nm, float(g['dt']), zmin=zmin,
                     n_order=n_order, particle_shape=shape,
                     n_guard=(None if n_order == -1 else 8), **extra)
    sim.ptcl = []
    nsp = len(g['q'])
    for isp in range(nsp):
        s = sim.add_new_species(q=float(g['q'][isp]), m=float(g['m'][isp]))
        set_species_state(s, g['s0_ptcl%d' % isp])
        s.grid_shape = sim.grid_shape
    for m in range(Nm):
        for i, k in enumerate(INTERP):
            setattr(sim.fld.interp[m], k, g['s0_interp'][m, i].copy())
    return sim


