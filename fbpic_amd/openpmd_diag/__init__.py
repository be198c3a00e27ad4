"""Diagnostics hook without HDF5: the two hook points of
Simulation.step (main.py:478-482 after the gather, :564-565 at the end of an iteration) and
the copy-out convention of the reference's diagnostics, writing `.npz` files instead of
openPMD/HDF5 (h5py is not part of this build)."""
from .npz_diag import FieldDiagnostic, ParticleDiagnostic, Checkpoint
from .checkpoint_restart import set_periodic_checkpoint, restart_from_checkpoint

__all__ = ['FieldDiagnostic', 'ParticleDiagnostic', 'Checkpoint', 'set_periodic_checkpoint',
           'restart_from_checkpoint']
