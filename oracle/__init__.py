"""CPU oracle for the deflated-MLMC Schwinger trace path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported CPU baseline.
The product package ``deflatedmlmc_schwinger_amd`` never imports this package.
"""
