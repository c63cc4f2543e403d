"""The C ABI from C: tests/cabi/consumer.c is a plain C99 program (gcc -std=c99 -pedantic -Werror) on include/rtbhip.h and librtbhip.so -- what a
maintainer's cgo / JNI / N-API / ctypes binding sits on.  `-m "not gpu"`: the header is valid C, every entry point it declares links
against the library (no compute); `-m gpu`: the program builds the Panda, evaluates fkine + jacob0 on host rows and its checksum equals the oracle's."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "cabi")
LIBDIR = os.path.join(ROOT, "robotics-toolbox-python_amd", "lib")


def declared():
    text = open(os.path.join(ROOT, "include", "rtbhip.h")).read()
    return re.findall(r"^(?:int|void|const char \*)\s*(rtbhip_\w+)\(", text, flags=re.M)


def build():
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler on this box")
    import __graft_entry__ as g
    g.build_lib()
    os.makedirs(OUT, exist_ok=True)
    names = declared()
    open(os.path.join(OUT, "symbols.inc"), "w").write("".join("X(%s)\n" % n for n in names))
    exe = os.path.join(OUT, "consumer")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + OUT,
                           os.path.join(ROOT, "tests", "cabi", "consumer.c"), "-o", exe, "-L" + LIBDIR, "-lrtbhip", "-lm", "-Wl,-rpath," + LIBDIR])
    return exe, names


def test_header_is_c99_and_every_declared_entry_point_links():
    exe, names = build()
    from rtbhip import _lib
    assert len(names) == len(set(names)) == len(_lib.SIGNATURES) and set(names) == set(_lib.SIGNATURES)      # the ctypes table names the same functions
    out = subprocess.check_output([exe, "symbols"], text=True)
    assert out.split() == ["symbols", str(len(names)), "version", "100"]


@pytest.mark.gpu
def test_gpu_c_program_computes_what_the_oracle_computes():
    from oracle import oracle, chains
    exe, _ = build()
    N = 1000
    out = subprocess.check_output([exe, "fkine", str(N)], text=True).split()
    assert out[:6] == ["joints", "7", "ets", "22", "rows", str(N)]
    ch = chains.Chain(chains.PANDA_ETS)
    q = 0.1 * (np.arange(7) + 1)[None, :] + 1e-3 * np.arange(N)[:, None]
    T, J = oracle.fkine(ch, q), oracle.jacob0(ch, q)
    want = float((T.reshape(N, 16) * (1 + np.arange(16))).sum() + (J.reshape(N, 42) * (1 + np.arange(42))).sum())
    assert abs(float(out[7]) - want) < 1e-7 * max(1.0, abs(want))


@pytest.mark.gpu
@pytest.mark.parametrize("p2p", [0, 1])
def test_gpu_c_program_shards_and_gathers_over_rccl(p2p):
    """`consumer shard N`: one C process, one RCCL communicator + stream per visible GPU (world 1 on a single-GPU box), packed T||J rows computed
    per rank, ONE rtbhip_shard_gather to rank 0 and one to every rank -- no PyTorch anywhere in the process.  p2p = 1 forces the grouped
    send / receive form ragged shards take."""
    from oracle import oracle, chains
    exe, _ = build()
    N = 1003
    r = subprocess.run([exe, "shard", str(N)] + (["p2p"] if p2p else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.startswith("world ")][-1].split()
    d = dict(zip(out[0::2], out[1::2]))
    import torch
    assert int(d["world"]) == torch.cuda.device_count() == int(d["comm_world"]) and int(d["comm_rank"]) == int(d["world"]) - 1
    assert int(d["rccl"]) > 20000 and int(d["rows"]) == N and d["allgather_equal"] == "1" and int(d["p2p"]) == p2p
    ch = chains.Chain(chains.PANDA_ETS)
    q = 0.1 * (np.arange(7) + 1)[None, :] + 1e-3 * np.arange(N)[:, None]
    T, J = oracle.fkine(ch, q), oracle.jacob0(ch, q)
    want = float((T.reshape(N, 16) * (1 + np.arange(16))).sum() + (J.reshape(N, 42) * (1 + np.arange(42))).sum())
    assert abs(float(d["checksum"]) - want) < 1e-7 * max(1.0, abs(want))
