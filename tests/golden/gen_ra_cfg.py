#!/usr/bin/env python3
"""Writes tests/golden/bitstreams/test_ra_gop16.cfg: this repository's own encoder configuration carrying the parameter VALUES of the
reference's random-access configuration (GOP structure table, intra period, search range / ASR, tool switches).  The values were read from
the reference's cfg (study); the file, its layout and comments are this repository's.  Kept for provenance: the table below IS the cfg."""
print(open(__file__.replace("gen_ra_cfg.py", "bitstreams/test_ra_gop16.cfg")).read())
