"""CPU oracle of the GraphSAGE hot path -- TEST INFRASTRUCTURE ONLY (see oracle/cpu.py header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
