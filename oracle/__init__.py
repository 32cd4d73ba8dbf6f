"""TEST INFRASTRUCTURE ONLY.

`oracle/` holds the CPU checker for the GLAMR global-reconstruction hot path: a restatement of the third-party
smplx arithmetic (`smplx_lbs.py`), a CPU port of the reference path (`port/`), and the harness that runs the
*unmodified* reference under import stubs in the build container (`ref_harness.py`, needs /root/reference).

Nothing under glamr_amd/ may import from here.  Allowed importers: tests/, __graft_entry__.smoke(), and the
`cpu_baseline` leg of bench.py.
"""
