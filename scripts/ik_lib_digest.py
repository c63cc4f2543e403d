#!/usr/bin/env python3
"""scripts/ik_lib_digest.py FILE -- per library: the sustained ms of every round for each workload of scripts/ik_lib_time.py, and whether the
(success, iterations, searches) checksums equal the first library's."""
import json
import sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
libs = []
for r in rows:
    name = r["lib"].split("/")[-1] or "product"
    if name not in libs:
        libs.append(name)
ref = None
for name in libs:
    mine = [r for r in rows if (r["lib"].split("/")[-1] or "product") == name]
    parts = []
    for wl in ("config3", "notebook", "1e6"):
        ms = [m[wl]["ms"] for m in mine]
        parts.append("%s %s (mean %.4f)" % (wl, " ".join("%.4f" % x for x in ms), sum(ms) / len(ms)))
    sha = tuple(mine[0][wl]["counts_sha"] for wl in ("config3", "notebook", "1e6"))
    bits = tuple(mine[0][wl].get("bits_sha") for wl in ("config3", "notebook", "1e6"))
    ref = ref or (sha, bits)
    print("%-28s %s | counts %s | q, E bits %s" % (name, " | ".join(parts), "equal" if sha == ref[0] else "DIFFER %r vs %r" % (sha, ref[0]),
                                                   "equal" if bits == ref[1] else "differ"))
