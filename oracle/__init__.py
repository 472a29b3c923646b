"""CPU oracle for the NTT / RNS hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package, and only as the checker.  The product
(``fhe_reliability_gpu_amd``) never imports it and has no CPU fallback.

Two restatements live here:

* ``oracle.cport``  -- ctypes wrapper over ``fhe_oracle.c`` (plain C, u128), fast
  enough for N = 2^16/2^17 checks;
* ``oracle.pyport`` -- pure-Python (big-int) restatement that follows the
  reference's Python line by line; small cases and the ``cpu_python`` baseline.

Pinning status is recorded in the header of ``fhe_oracle.c`` and in DESIGN.md.
"""
