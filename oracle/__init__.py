"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/osqp_ref.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
