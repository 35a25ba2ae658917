"""ORACLE — test infrastructure only.

CPU restatement of the GLASS inference hot path used as the parity checker.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from here; the
product (glass-text-spotting_amd/) never does, and fails loudly without its HIP library.
"""
