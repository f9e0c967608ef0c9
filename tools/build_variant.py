#!/usr/bin/env python3
"""Builds a variant of libpphip.so for tuning experiments:  tools/build_variant.py NAME -DFLAG=1 ...
-> pathplanning_amd/lib/variants/NAME.so ; select it at run time with PP_HIP_LIB=<that path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathplanning_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(B.LIB_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
out = os.path.join(out_dir, name + ".so")
subprocess.check_call([B.hipcc()] + B.FLAGS + flags + ["-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES], cwd=B.CSRC)
print(out)
