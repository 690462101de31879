#!/usr/bin/env python3
"""CLI of the report driver (gpu-wah_amd/report.py), the counterpart of the reference's main() in source.cpp:
    python tools/report.py                      # the reference's full sweep: 9 sizes x 16 densities x 10 repetitions
    python tools/report.py --max-s 4 --densities 1,4,8 --reps 3 --out small.csv
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if __name__ == "__main__":
    sys.exit(importlib.import_module("gpu-wah_amd.report").main())
