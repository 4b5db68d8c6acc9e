#!/usr/bin/env python3
"""Copy the reference's example driver examples/run_karate.py into tests/golden/ as a TEST INPUT (byte-identical, never
imported by the product): tests/test_run_karate_gpu.py executes it unchanged against the `gem` alias package on the GPU box,
where /root/reference does not exist.  tests/test_run_karate_cpu.py checks here that the copy is still byte-identical."""
import os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shutil.copyfile('/root/reference/examples/run_karate.py', os.path.join(ROOT, 'tests', 'golden', 'ref_examples_run_karate.py.txt'))
print('ok')
