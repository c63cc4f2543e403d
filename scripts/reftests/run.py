"""Ad-hoc: run one of the reference's test files against rtbhip (exploration tool behind tests/test_reference_suite.py)."""
import sys, os, unittest, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "robotics-toolbox-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_reference_suite as T
from oracle import ref_classes
name = sys.argv[1]
if name + ".py" not in ref_classes.TEST_FILES:
    ref_classes.TEST_FILES.append(name + ".py")
saved = T.install_shims()
try:
    res = T.run_module(ref_classes.load_test_module(name))
finally:
    T.restore(saved)
bad = {k: v for k, v in res.items() if v is not None}
print(name, len(res), "tests,", len(res) - len(bad), "pass")
for k, v in sorted(bad.items()): print("  ", k, "::", v[:200])
