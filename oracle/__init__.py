"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference hot path (parity checkers).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
