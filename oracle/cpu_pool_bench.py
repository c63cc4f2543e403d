#!/usr/bin/env python3
"""oracle/cpu_pool_bench.py -- CPU BASELINE HELPER (TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE).

The reference's fkine + per-row jacob0 path (oracle/_ref, its own extension built unmodified) on every
host core: the extension holds the GIL and has no threads, so one process per core over row blocks
(SURVEY 8d ii).  Started by bench.py's cpu_baseline leg as a separate, time-limited process; prints one
JSON object.  Never imports the product or the HIP runtime.

    python oracle/cpu_pool_bench.py q.npy
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_REF = {}


def _init():
    from oracle import chains, ref_harness
    _REF["ets"] = ref_harness.RefETS(chains.panda_ets())


def _work(block):
    ref = _REF["ets"]
    ref.fkine(block)
    ref.jacob0_batch(block)
    return len(block)


def main():
    sample = np.load(sys.argv[1])
    cores = os.cpu_count() or 1
    blocks = np.array_split(sample, cores * 4)
    with mp.get_context("fork").Pool(cores, initializer=_init) as pool:
        pool.map(_work, [b[:64] for b in blocks])          # start-up is not part of the measurement
        best, used = None, 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            pool.map(_work, blocks)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            used += dt
            if used > 20.0:
                break
    print(json.dumps({"value": len(sample) / best, "unit": "configurations/s", "cores": cores,
                      "sample": "the same %d configurations, %d row blocks over a %d-process pool, best of up to 3 passes"
                                % (len(sample), len(blocks), cores)}))


if __name__ == "__main__":
    main()
