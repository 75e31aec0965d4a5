"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on the PointSegment hot path, plus ctypes doors into
the real reference C++ compiled under oracle/_ref (when present).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product package never does.
"""
