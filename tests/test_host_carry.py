"""Host-side logic of the state carried across Simulation.step calls (no GPU): the deferred
J / rho attributes of the interpolation grids (Fields.defer_sources /
InterpolationGrid.__getattr__), the deferred E, B attributes of a species, and when a carry
signature exists at all.  The numerical side is tests/test_gpu_carry.py."""
import numpy as np
import pytest
import torch
import helpers


def _small_sim():
    return helpers.uniform_plasma_sim(8, 4, 2, (1, 1, 4), 'linear', seed=1)


def test_deferred_sources_are_brought_back_on_first_read_only():
    sim = _small_sim()
    fld = sim.fld
    Nm = fld.Nm
    # stand-in for the device slab (the mechanism is tensor-type agnostic)
    fld.d_interp = torch.zeros((fld.Nz, 10 * Nm, fld.Nr), dtype=torch.complex128)
    calls = []

    def bring_back():
        calls.append(1)
        fld.d_interp[:, fld.interp_index('Jz', 1), :] = 7.
        fld.d_interp[:, fld.interp_index('rho', 0), :] = 3.
    fld.defer_sources(bring_back)
    for g in fld.interp:
        assert not any(k in g.__dict__ for k in ('Jr', 'Jt', 'Jz', 'rho'))
        assert 'Er' in g.__dict__                       # E, B are never deferred
    assert calls == []
    assert float(fld.interp[1].Jz.real.max()) == 7. and calls == [1]      # first read: computed
    assert float(fld.interp[0].rho.real.max()) == 3. and calls == [1]     # ... once, for all four
    assert fld._deferred_sources is None
    fld.materialize_sources()
    assert calls == [1]
    # dropped without being computed (what the next step() call does)
    fld.defer_sources(bring_back)
    fld.drop_deferred_sources()
    assert calls == [1] and 'Jr' in fld.interp[0].__dict__
    # an attribute that does not exist still raises
    with pytest.raises(AttributeError):
        fld.interp[0].no_such_field
    # a direct erase of the sources (a deposit outside step) first brings the others back
    fld.defer_sources(bring_back)
    fld.data_is_on_gpu = True
    try:
        with pytest.raises(Exception):
            fld.erase('rho')            # no GPU here: the launch itself fails ...
    finally:
        fld.data_is_on_gpu = False
    assert calls == [1, 1]              # ... after the deferred sources were materialised


def test_deferred_particle_fields_can_be_dropped():
    sim = _small_sim()
    sp = sim.ptcl[0]
    store = [getattr(sp, k) for k in ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz')]
    sp._field_store = [sp.__dict__.pop(k) for k in ('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz')]
    sp._deferred_fields = ('views', 2, 1., (1., 0., 8, 1., 0., 4), 1e-16)
    assert 'Ex' not in sp.__dict__
    sp.drop_deferred_fields()
    assert sp._deferred_fields is None and all(getattr(sp, k) is a for k, a in
                                               zip(('Ex', 'Ey', 'Ez', 'Bx', 'By', 'Bz'), store))
    with pytest.raises(AttributeError):
        sp.no_such_attribute


def test_no_carry_signature_without_resident_data_or_for_excluded_schemes():
    sim = _small_sim()
    assert sim._carry_signature() is None               # data on the host
    sim.carry_state_between_calls = False
    assert sim._carry_signature() is None
    assert sim._can_defer_particle_fields() is False


def test_version_counter_self_test(monkeypatch):
    """The carried state depends on torch's private Tensor._version: checked once per process,
    and nothing is carried where the check fails."""
    from fbpic_amd import main
    assert main._version_counter_works() is True        # this torch build
    monkeypatch.setattr(main, '_VERSION_COUNTER_OK', [False])
    sim = _small_sim()
    sim.fld.data_is_on_gpu = True
    for s in sim.ptcl:
        s.data_is_on_gpu = True
    assert sim._carry_signature() is None
