"""What the BUILT library's structure instantiations look like to the hardware (no GPU needed: the gfx950 code objects inside librtbhip.so are read
with llvm-objdump / llvm-readelf of the ROCm image).

Round 6 found the built-in link-tree instantiations (UR, the Interbotix arms, Fetch, Mico: 28 kernels) compiled as ROLLED loops after the recursions'
products had been written out -- the `#pragma unroll` bodies had grown past the compiler's default cap, a group's class lookups were then decoded at run
time and its state lived in scratch: the UR5's "fast" kernel ran 2.8x slower than the general one, and only the driver's bench line showed it.  The build
(and hipRTC: csrc/jit.cpp) now passes -pragma-unroll-threshold; this test pins what that buys: a structure instantiation is straight-line code that keeps
its state in registers -- no private (scratch) segment, no spilled vector registers."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "robotics-toolbox-python_amd", "lib", "librtbhip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(tmp):
    """{mangled kernel name: {note: int}} over every gfx950 code object bundled in the library"""
    shutil.copy(LIB, os.path.join(tmp, "lib.so"))
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", e).group(1)
            out[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1)) for k in
                         ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return out


@pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))), reason="ROCm llvm tools not found")
def test_structure_instantiations_are_straight_line_register_code(tmp_path):
    import __graft_entry__ as g
    g.build_lib()
    ks = _kernels(str(tmp_path))
    assert len(ks) > 500, len(ks)
    # the built-in structure instantiations: link trees with a non-zero first knowledge word, k_ik / k_rne / k_rne_atrest / k_dyn with a non-zero signature
    tree = {n: v for n, v in ks.items() if re.search(r"k_tree_(rne|dyn)I.*TreeKnownILy[1-9]", n)}
    ik = {n: v for n, v in ks.items() if re.search(r"4k_ikILi\d+ELi0ELi\d+ELy[1-9]", n)}
    rne = {n: v for n, v in ks.items() if re.search(r"(5k_rne|12k_rne_atrest|5k_dyn)I.*ELy[1-9]\d+EEE", n)}
    assert len(tree) >= 28 and len(ik) >= 6 and len(rne) >= 4, (len(tree), len(ik), len(rne))
    bad = {n: v for n, v in {**tree, **ik, **rne}.items() if v["private_segment_fixed_size"] or v["vgpr_spill_count"]}
    assert not bad, "structure instantiations with scratch: %r" % ({n[:70]: v for n, v in list(bad.items())[:4]},)
    # the kernel that serves config 3 keeps two waves per SIMD and eight waves per CU of LDS (profiles/r06_ik_three_waves.txt)
    panda = [v for n, v in ik.items() if "ELy9265531810339127745E" in n and re.search(r"ELi0ELi1[23]ELy", n)]
    assert panda and all(v["vgpr_count"] <= 256 and v["group_segment_fixed_size"] * 8 <= 160 * 1024 for v in panda), panda
