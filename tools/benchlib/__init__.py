"""The pieces of bench.py: kernels (per-kernel timings and rooflines), legs (the timed workloads), launch (ranks, preflight, scaling), line (the one stdout line)."""
