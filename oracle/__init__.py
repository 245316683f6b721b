"""TEST INFRASTRUCTURE.  Python bindings of the oracle libraries.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this package.  Nothing under raytracer_amd/ does.
"""
