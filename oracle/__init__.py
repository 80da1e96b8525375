"""CPU oracle for the TrackNetV3 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product package (``tracknetv3_amd``) never imports this package and fails
loudly when its HIP library is missing.
"""
