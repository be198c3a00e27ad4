"""fbpic_amd.lpa_utils: part of the MI355X (gfx950) backend of the FBPIC per-step PIC cycle."""
