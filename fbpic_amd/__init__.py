"""fbpic_amd: MI355X (gfx950) backend for FBPIC's per-step PIC cycle.

Same Python surface as the reference for the hot path (`Simulation.step`, `Particles`,
`Fields`), executed by hand-written HIP kernels in csrc/ through a C ABI
(include/fbpic_amd.h).  There is no CPU execution path.
"""
__version__ = '0.1.0'
