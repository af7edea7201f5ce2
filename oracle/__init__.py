"""CPU oracle for the RTM3D inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``rtm3d_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / the timed CPU baseline, never as the product path.

Parity pin: the functions here are checked against golden vectors that were produced by
importing the real reference (``/root/reference``) in the build container with
``tests/golden/make_golden.py`` (see ``tests/test_oracle_golden.py``).
"""
