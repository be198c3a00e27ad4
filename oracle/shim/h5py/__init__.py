"""Empty stand-in: only the reference diagnostics (off the hot path) need real h5py."""
