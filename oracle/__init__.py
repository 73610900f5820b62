"""CPU oracle for the sketch->(RGB,tactile) GAN hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under visual-tactile-synthesis_amd/ imports this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do.
"""
