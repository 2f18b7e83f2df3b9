#!/usr/bin/env python3
"""Longer soak of tests/test_gpu_parity.py::test_random_worlds_against_oracle: random worlds, cameras, regions, sample counts
and bounce limits; the GPU in all four precisions (and the f32 mode through the hierarchy and LDS) against the CPU oracle,
bitwise.   usage: tools/soak_random_worlds.py [first_case] [cases]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as t
import oracle_lib
rt = importlib.import_module("raytracing-in-one-weekend_amd")
oracle = oracle_lib.Oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for case in range(first, first + cases):
    t.test_random_worlds_against_oracle(rt, oracle, case)
    if (case - first) % 20 == 19:
        print("cases done:", case - first + 1, flush=True)
print("soak ok:", cases, "random worlds")
