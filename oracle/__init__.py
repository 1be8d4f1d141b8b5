"""Parity oracle for the Co-Occ fused-voxel hot path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (co_occ_amd/) never imports it.
"""
