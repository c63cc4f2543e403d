#!/usr/bin/env python3
"""scripts/aot_manifest.py -- fold RTBHIP_JIT_MANIFEST logs into robotics-toolbox-python_amd/jit_aot_manifest.jsonl.

    RTBHIP_JIT_MANIFEST=$PWD/gpurun_out/<visit>/jit_manifest.jsonl  <the suite, the benches, the fuzzers on the device>
    python scripts/aot_manifest.py gpurun_out/<visit>/jit_manifest.jsonl [more logs ...]

Every run-time instantiation a process asks for (csrc/jit.cpp: manifest_note) is one JSON line {"unit", "expr", "preamble"}.  The committed list is
those lines without repeats, sorted -- what __graft_entry__.build_aot_cache compiles ahead of time into lib/jitcache/.  With --replace the committed
list is rebuilt from the logs alone; otherwise the logs are merged into it."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "robotics-toolbox-python_amd", "jit_aot_manifest.jsonl")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
seen = {}
srcs = list(args) + ([] if "--replace" in sys.argv or not os.path.exists(OUT) else [OUT])
bad = 0
for path in srcs:
    for line in open(path, errors="replace"):
        line = line.strip()
        if not line:
            continue
        try:
            e = json.loads(line)
            key = (e["unit"], e["expr"], e["preamble"])
        except (ValueError, KeyError):
            bad += 1
            continue
        seen[key] = {"unit": e["unit"], "expr": e["expr"], "preamble": e["preamble"]}
with open(OUT, "w") as f:
    for key in sorted(seen):
        f.write(json.dumps(seen[key], sort_keys=True) + "\n")
print("%d instantiations -> %s (%d unreadable lines skipped, %d bytes)" % (len(seen), os.path.relpath(OUT, ROOT), bad, os.path.getsize(OUT)))
