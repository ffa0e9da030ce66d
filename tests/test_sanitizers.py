"""The CPU-side C under AddressSanitizer + UndefinedBehaviorSanitizer (CPU builds only -- no GPU sanitizer, no XNACK).

The oracle is the only judge of the kernels, so it gets a judge of its own: `make -C oracle SAN=1` builds the same four sources with
-fsanitize=address,undefined -fno-sanitize-recover=all, and the oracle's own test files (golden fixtures, oracle == independent model on random
inputs and on the full prime chains, BFV, key generation, the reference's property tests) run against THAT library in a child interpreter
(libasan has to be the first library of the process: LD_PRELOAD; MKHE_ORACLE_LIB selects the build).  Any out-of-bounds access of the fixed scratch
arrays (y[32] / sp[32], ora_mkrlwe.c), misaligned or overflowing arithmetic, or use after free aborts the child and fails this test.
The C++ host mirror (include/mkhe.hpp through tests/cpp/mirror_check.cpp) is compiled with UBSan here and RUN with it by the GPU test
tests/test_cpp_mirror.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_LIB = os.path.join(ROOT, "oracle", "_build", "san", "libmkhe_oracle.so")
FILES = ["test_golden.py", "test_oracle_vs_model.py", "test_bfv_oracle.py", "test_keygen_oracle.py", "test_oracle_full_chains.py", "test_reference_properties.py"]


def _gcc_file(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True, check=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_asan_and_ubsan():
    asan, stdcxx = _gcc_file("libasan.so"), _gcc_file("libstdc++.so.6")
    if not asan:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "SAN=1"])
    assert os.path.exists(SAN_LIB)
    sym = subprocess.run(["nm", "-D", SAN_LIB], capture_output=True, text=True, check=True).stdout
    assert "__asan_init" in sym and "__ubsan_handle" in sym, "the SAN=1 build is not instrumented"
    env = dict(os.environ, MKHE_ORACLE_LIB=SAN_LIB, LD_PRELOAD=" ".join(p for p in (asan, stdcxx) if p),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import oracle as O; print(O.lib()._name)" % ROOT],
                           env=env, capture_output=True, text=True, timeout=120)
    assert probe.returncode == 0 and probe.stdout.strip() == SAN_LIB, (probe.stdout, probe.stderr[-2000:])
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in FILES],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    assert " passed" in r.stdout


def test_cpp_mirror_builds_under_ubsan(tmp_path):
    """include/mkhe.hpp + tests/cpp/mirror_check.cpp with -fsanitize=undefined -fno-sanitize-recover (the run is the GPU test test_cpp_mirror.py)"""
    exe = str(tmp_path / "mirror_check_ubsan")
    from oracle import oracle as O
    O.build()
    lib, ora = os.path.join(ROOT, "mkhe-kklss_amd", "lib"), os.path.join(ROOT, "oracle", "_build")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp"),
                           "-o", exe, "-L", lib, "-lmkhe_hip", "-L", ora, "-lmkhe_oracle", "-Wl,-rpath," + lib, "-Wl,-rpath," + ora, "-Wl,--allow-shlib-undefined"])
    assert os.path.exists(exe)
