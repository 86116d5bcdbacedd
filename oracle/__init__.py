"""CPU oracle of the `pt` hot path: TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package or load the
shared libraries built from it. The shipped HIP path (akari_render_amd) never does.
"""
