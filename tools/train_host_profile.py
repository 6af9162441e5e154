#!/usr/bin/env python3
"""Diagnostic: cProfile of the host side of 12 optimizer steps (tools/train_step_time.measure) with HipSGD or torch's SGD."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_step_time as TS
hip = "--torch-sgd" not in sys.argv
pr = cProfile.Profile()
pr.enable()
TS.measure(4096, True, 12, hip_sgd=hip)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(s.getvalue().splitlines()[:45]))
