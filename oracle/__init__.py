"""TEST INFRASTRUCTURE ONLY: CPU oracle for the DDSP hot path (see ddsp_oracle.c).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
