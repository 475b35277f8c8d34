"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package ``gpuspectral_amd`` never imports it.
"""
