"""The "Use" block of README.md, executed as it is printed (at a reduced batch size): documentation that cannot rot."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def readme_block():
    text = open(os.path.join(ROOT, "README.md")).read()
    m = re.search(r"## Use.*?```python\n(.*?)```", text, re.S)
    assert m, "README.md has no python block under '## Use'"
    return m.group(1)


def test_readme_block_is_python():
    compile(readme_block(), "README.md", "exec")


@pytest.mark.gpu
def test_readme_block_runs_on_the_gpu():
    src = readme_block().replace("1_000_000", "20_000").replace("100_000", "2_000")
    assert "20_000" in src
    import sys
    saved = {k: sys.modules.get(k) for k in ("roboticstoolbox.fknm", "roboticstoolbox.frne")}       # the last line installs the shims
    ns = {}
    try:
        exec(compile(src, "README.md", "exec"), ns)                # noqa: S102 -- our own README
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert ns["T"].shape[-2:] == (4, 4) and ns["tau"].shape[1] == 6 and ns["wbase"].shape[1] == 6
