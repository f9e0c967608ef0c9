#!/usr/bin/env python3
"""Collision-check throughput only (the `secondary` block of bench.py): python tools/bench_checks.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--streams", "1", "--batch", "64", "--no-cpu-baseline"],
                     capture_output=True, text=True)
for line in out.stdout.splitlines():
    if line.startswith("{"):
        print(json.dumps(json.loads(line)["secondary"]))
        break
else:
    print(out.stdout[-2000:], out.stderr[-2000:])
    sys.exit(1)
