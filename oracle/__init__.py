"""CPU oracle for the paint-coverage simulator -- TEST INFRASTRUCTURE ONLY.

See paint_oracle.c for the contract.  Importable from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg; never from
paintrl_amd/.
"""
from .paint_oracle import Oracle, build, lib_path  # noqa: F401
