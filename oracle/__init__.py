"""CPU oracle for the CMFlow hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package; nothing under ``cmflow_amd/`` does (tests/test_layout.py checks).
"""
