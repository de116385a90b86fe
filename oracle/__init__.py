"""ORACLE -- test infrastructure only.

CPU restatements of the reference's algorithm for the hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import anything from this package; the product (eigensolver_amd/) never does.
"""
