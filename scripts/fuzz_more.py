#!/usr/bin/env python3
"""scripts/fuzz_more.py OFFSET [OFFSET ...] -- the device fuzzers of scripts/gpu_fuzz_*.py over OTHER random robots than the ones the test suite
visits: every integer seed handed to numpy.random.default_rng is shifted by OFFSET (the fuzzers themselves are untouched), each family in a process
of its own.  One line per (offset, family): exit code and the fuzzer's last line; exit code 1 if any family missed."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRELUDE = r"""
import sys, runpy, numpy as np
_rng, OFF = np.random.default_rng, %d
np.random.default_rng = lambda seed=None, *a, **k: _rng(seed + OFF if isinstance(seed, (int, np.integer)) else seed, *a, **k)
sys.argv = [%r]
runpy.run_path(%r, run_name="__main__")
"""
bad = 0
for off in [int(x) for x in sys.argv[1:]] or [100000]:
    for fam in ("kin", "rne", "dyn", "ik", "paths", "fleet", "jit"):
        path = os.path.join(ROOT, "scripts", "gpu_fuzz_%s.py" % fam)
        r = subprocess.run([sys.executable, "-c", PRELUDE % (off, path, path)], capture_output=True, text=True, timeout=1500, cwd=ROOT)
        last = (r.stdout.strip().splitlines() or [""])[-1]
        print(json.dumps({"offset": off, "family": fam, "rc": r.returncode, "last": last[:600], "err": r.stderr.strip()[-300:] if r.returncode else ""}), flush=True)
        bad += r.returncode != 0
sys.exit(1 if bad else 0)
