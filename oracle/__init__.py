"""TEST INFRASTRUCTURE ONLY.  CPU oracles for the HIP kernels (see each module's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package;
nothing under video-layout-generation_amd/ does.
"""
