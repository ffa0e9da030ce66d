"""The C++ host mirror include/mkhe.hpp (same type / method names as the Go packages mkrlwe, mkckks, mkbfv): it must
compile and link against the C ABI on CPU, and -- on the GPU -- its results must equal the oracle's bit for bit
(tests/cpp/mirror_check.cpp links the oracle as the checker)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mkhe-kklss_amd", "lib")
ORA = os.path.join(ROOT, "oracle", "_build")


def build(tmp_path):
    from oracle import oracle as O
    O.build()
    exe = str(tmp_path / "mirror_check")
    # UBSan on the host mirror and its checker: a misaligned access, a shift or a signed overflow in include/mkhe.hpp aborts the run
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp"), "-o", exe,
                           "-L", LIB, "-lmkhe_hip", "-L", ORA, "-lmkhe_oracle",
                           "-Wl,-rpath," + LIB, "-Wl,-rpath," + ORA, "-Wl,--allow-shlib-undefined"])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_cpp_mirror_matches_oracle(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "all C++ mirror checks passed" in out.stdout
