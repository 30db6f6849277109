"""CPU oracle for the hot path.  TEST INFRASTRUCTURE ONLY — see fmgan_oracle.c / torch_oracle.py headers.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
