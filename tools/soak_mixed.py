#!/usr/bin/env python3
"""Longer soak of tests/test_gpu_parity.py::test_mixed_equals_f64_kernel_on_adversarial_worlds: random worlds from
millimetres to kilometres, cameras up to a thousand world sizes away; the mixed image (list scan and hierarchy) must be
the f64 kernel's.   usage: tools/soak_mixed.py [first_block] [blocks]   (10 worlds per block)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as t
rt = importlib.import_module("raytracing-in-one-weekend_amd")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for block in range(first, first + blocks):
    t.test_mixed_equals_f64_kernel_on_adversarial_worlds(rt, block)
    if (block - first) % 10 == 9:
        print("blocks done:", block - first + 1, flush=True)
print("soak ok:", 10 * blocks, "worlds")
