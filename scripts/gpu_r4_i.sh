#!/bin/bash
# Round 4, visit i: A/B of IK builds (RTBHIP_LIB): LDL with kept unnormalised entries (product) against the visit-h kernel (variant ik_noU); IK tests.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4i}
mkdir -p $O
cd $R
V=$R/robotics-toolbox-python_amd/lib/variants
for round in 1 2 3; do
  for lib in "" $(ls $V/*.so); do
    RTBHIP_LIB=$lib timeout 300 python scripts/ik_lib_time.py 2>/dev/null >> $O/ik_lib_ab.jsonl
  done
done
python - $O/ik_lib_ab.jsonl <<'PY'
import json, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
info = {}
for l in open(sys.argv[1]):
    d = json.loads(l)
    name = d["lib"].split("/")[-1] or "product"
    for k in ("config3", "notebook", "1e6"):
        rows[name][k].append(d[k]["ms"]); info[(name, k)] = (d[k]["ok"], d[k]["its"], d[k]["counts_sha"])
for name, r in rows.items():
    print("%-18s" % name, {k: v for k, v in r.items()})
for k in ("config3", "notebook", "1e6"):
    print(k, {n: info[(n, k)] for n in rows})
PY
timeout 900 python -m pytest tests/test_00_gpu_parity.py tests/test_03_python_ik_pins.py tests/test_02_compat_shim.py tests/test_05_reference_classes.py -m gpu -q -rf --timeout 600 2>&1 | tail -5
