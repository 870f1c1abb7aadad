"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the MonoDETR / MSDeformAttn hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package; the product (``monosowa_amd``) never does.
"""
