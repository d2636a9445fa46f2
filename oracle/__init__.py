"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's algorithms for the nnUZoo hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package - and only as the checker, never as the thing shipped or
measured.  The product (nnuzoo_amd/) never imports it.
"""
