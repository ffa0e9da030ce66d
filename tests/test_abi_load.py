"""CPU: the C-ABI shared library loads and exports every symbol that include/mkhe.h declares, and the
ctypes binding covers the same set (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mkhe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mkhe_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for must in ("mkhe_ctx_create", "mkhe_decompose", "mkhe_external_product", "mkhe_external_product_hoisted",
                 "mkhe_mul_and_relin", "mkhe_rotate", "mkhe_conjugate", "mkhe_rescale", "mkhe_ntt"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from mkhe_kklss_amd import _abi
    assert os.path.exists(_abi.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = ctypes.CDLL(_abi.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), "libmkhe_hip.so does not export %s" % s


def test_binding_matches_header():
    from mkhe_kklss_amd import _abi
    assert sorted(_abi.SIGNATURES) == declared_symbols()
    _abi.lib()          # resolves every symbol with its signature


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mkhe_kklss_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _abi.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mkhe-kklss_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), "%s mentions the oracle" % f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """cgo needs a C header: include/mkhe.h must compile as C99, and a C translation unit that references every
    declared entry point must link against libmkhe_hip.so (nothing is called: there is no GPU here)."""
    import subprocess
    from mkhe_kklss_amd import _abi
    hdr = os.path.join(ROOT, "include", "mkhe.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    src = tmp_path / "use.c"
    body = "\n".join("    p[%d] = (void*)%s;" % (i, s) for i, s in enumerate(declared_symbols()))
    src.write_text('#include "mkhe.h"\n#include <stdio.h>\nint main(void) {\n    void* p[%d];\n%s\n'
                   '    printf("%%d entry points\\n", (int)(sizeof p / sizeof p[0]));\n    return p[0] == 0;\n}\n'
                   % (len(declared_symbols()), body))
    exe = tmp_path / "use"
    libdir = os.path.dirname(_abi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lmkhe_hip", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"])
    assert os.path.exists(exe)


def test_product_library_reads_only_the_documented_environment():
    """A/B switches are compiled into libmkhe_hip_switches.so only.  Every MKHE_* string of the product library is a
    configuration variable that include/mkhe.h documents under "Environment", and there are at most ten of them."""
    import re
    import subprocess
    from mkhe_kklss_amd import _abi
    out = subprocess.run(["strings", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted({l.strip() for l in out.splitlines() if re.fullmatch(r"MKHE_[A-Z0-9_]+", l.strip())})
    hdr = open(os.path.join(ROOT, "include", "mkhe.h")).read()
    env_doc = hdr[hdr.index("Environment (the COMPLETE list"):hdr.index("#ifndef MKHE_H")]
    assert 0 < len(names) <= 10, names
    for n in names:
        assert n in env_doc, "%s is read by the product library but not documented in include/mkhe.h" % n
    sw = os.path.join(os.path.dirname(_abi.LIB_PATH), "libmkhe_hip_switches.so")
    assert os.path.exists(sw), "build() makes the -DMKHE_SWITCHES library too"
    out = subprocess.run(["strings", sw], capture_output=True, text=True, check=True).stdout
    assert sum(1 for l in out.splitlines() if re.fullmatch(r"MKHE_[A-Z0-9_]+", l.strip())) >= 25          # (round 6: 27 switch names left after the pruning)
