"""fbpic_amd.particles: part of the MI355X (gfx950) backend of the FBPIC per-step PIC cycle."""
