"""The benchmark's and the tests' world -- NOT part of the product package.

synthetic.py   deterministic inputs: sensor depth frames, the procedural PR2-like robot
workloads.py   BASELINE.json configs C1-C3 as plain data fed through the product's host mirror
configs.py     per-GPU shares of configs C3 / C4 / C5 for bench.py and the multi-rank rehearsals

Imported by bench.py, tests/, scripts/ and __graft_entry__.smoke(); realtime_urdf_filter_amd/ never imports it.
"""
