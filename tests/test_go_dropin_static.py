"""The signature-compatible Go drop-in (shim/go/dropin/**, un-built here: no Go toolchain) against the reference's own signatures.

tests/golden/ref_go_signatures.json is a table of the exported functions / methods of the reference files the drop-in stands in for (names,
parameter and result types; written by tools/gen_ref_go_signatures.py from /root/reference).  This test parses the drop-in with the same parser:

* package mkrlwe (keyswitch_gpu.go, build tag mkhe_gpu): `KeySwitcher` and `NewKeySwitcher` with EVERY exported method of mkrlwe/keyswitch.go and
  keyswitch_hoisted.go, identical parameter and result types -- the reference files are excluded by the tag (shim/go/patches), so a missing or
  changed method is a compile error in mkckks / mkbfv / the reference's tests; here it is a test failure;
* packages mkckks / mkbfv (evaluator_gpu.go): `GPUEvaluator` embeds *Evaluator (every method it does not redefine is the reference's, promoted) and
  each method it does redefine has the reference Evaluator's signature for that name; the key-switching methods must be redefined;
* the patches touch only what they say, and a self-test shows that a dropped parameter, a changed type and a missing method are reported."""
import copy
import glob
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_go_signatures.json")
DROPIN = os.path.join(ROOT, "shim", "go", "dropin")

_spec = importlib.util.spec_from_file_location("gen_ref_go_signatures", os.path.join(ROOT, "tools", "gen_ref_go_signatures.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


def golden():
    return json.load(open(GOLDEN))["packages"]


def dropin(pkg, tagged=True):
    """parsed signatures of the drop-in files of a package that are compiled WITH the mkhe_gpu tag"""
    funcs, methods, srcs = {}, {}, {}
    for f in sorted(glob.glob(os.path.join(DROPIN, pkg, "*.go"))):
        src = open(f).read()
        is_tagged = re.search(r"^//go:build mkhe_gpu$", src, flags=re.M) is not None
        assert is_tagged or re.search(r"^//go:build !mkhe_gpu$", src, flags=re.M), "%s: no build constraint" % f
        assert re.search(r"^package %s$" % pkg, src, flags=re.M), "%s is not a file of package %s" % (f, pkg)
        if is_tagged != tagged:
            continue
        p = gen.parse_go_signatures(src)
        funcs.update(p["funcs"])
        for t, ms in p["methods"].items():
            methods.setdefault(t, {}).update(ms)
        srcs[f] = src
    return funcs, methods, srcs


def types(sig_list):
    return [t for _, t in sig_list]


def compare(want, got, what):
    """-> list of problems"""
    if got is None:
        return ["%s: missing" % what]
    out = []
    if types(want["params"]) != types(got["params"]):
        out.append("%s: parameters %s, the reference has %s" % (what, types(got["params"]), types(want["params"])))
    if types(want["results"]) != types(got["results"]):
        out.append("%s: results %s, the reference has %s" % (what, types(got["results"]), types(want["results"])))
    return out


def check_keyswitcher(gold, funcs, methods):
    problems = compare(gold["mkrlwe"]["funcs"]["NewKeySwitcher"], funcs.get("NewKeySwitcher"), "mkrlwe.NewKeySwitcher")
    for name, sig in gold["mkrlwe"]["methods"]["KeySwitcher"].items():
        problems += compare(sig, methods.get("KeySwitcher", {}).get(name), "mkrlwe.KeySwitcher.%s (%s)" % (name, sig["file"]))
    return problems


HOT = {"mkckks": ["MulRelinNew", "MulRelinHoistedNew", "HoistedForm", "RotateNew", "RotateHoistedNew", "ConjugateNew", "RescaleNew"],
       "mkbfv": ["MulRelinNew"]}


def check_evaluator(pkg, gold, funcs, methods, srcs):
    problems = []
    ref = gold[pkg]["methods"]["Evaluator"]
    mine = methods.get("GPUEvaluator", {})
    if not any(re.search(r"type GPUEvaluator struct \{\s*\n\s*\*Evaluator\b", s) for s in srcs.values()):
        problems.append("%s.GPUEvaluator does not embed *Evaluator (the methods it leaves alone must be the reference's, promoted)" % pkg)
    for name in HOT[pkg]:
        if name not in mine:
            problems.append("%s.GPUEvaluator.%s: missing (a key-switching method must be an engine call)" % (pkg, name))
    for name, sig in mine.items():
        if name in ref:
            problems += compare(ref[name], sig, "%s.GPUEvaluator.%s (%s)" % (pkg, name, ref[name]["file"]))
    ctor = funcs.get("NewGPUEvaluator")
    problems += compare(dict(gold[pkg]["funcs"]["NewEvaluator"], results=[["", "*GPUEvaluator"]]), ctor, "%s.NewGPUEvaluator" % pkg)
    return problems


def test_golden_table_is_the_references(tmp_path):
    """the committed table equals what the generator reads from the reference (when the reference is mounted: the build container)"""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "mkrlwe")):
        pytest.skip("no reference checkout here")
    assert gen.build(ref)["packages"] == golden()


def test_table_holds_names_and_types_only():
    txt = open(GOLDEN).read()
    assert "{\n" in txt and "func " not in txt and ":=" not in txt and "ringQ." not in txt          # a table, not source text
    g = golden()
    assert set(g["mkrlwe"]["methods"]["KeySwitcher"]) == {"DecomposeSingleNTT", "Decompose", "ExternalProduct", "ExternalProductHoisted", "MulAndRelin",
                                                         "MulAndRelinHoisted", "Rotate", "RotateHoisted", "Conjugate"}
    assert len(g["mkckks"]["methods"]["Evaluator"]) == 14 and len(g["mkbfv"]["methods"]["Evaluator"]) == 5


def test_keyswitcher_dropin_has_the_references_method_set():
    funcs, methods, srcs = dropin("mkrlwe")
    assert not check_keyswitcher(golden(), funcs, methods), "\n".join(check_keyswitcher(golden(), funcs, methods))
    src = "\n".join(srcs.values())
    # mkbfv's host code reaches into the embedded lattigo key switcher and the Decomposer (mkbfv/keyswitch.go:92,110): they must stay fields
    assert re.search(r"type KeySwitcher struct \{\s*\n\s*rlwe\.KeySwitcher\s*\n\s*Parameters\s*\n\s*Decomposer \*Decomposer", src)
    for entry in ("g.Decompose(", "g.ExternalProduct(", "g.ExternalProductHoisted(", "g.MulAndRelinHoisted(", "g.RotateHoisted(", "g.Conjugate("):
        assert entry in src, "the drop-in never reaches %s" % entry


@pytest.mark.parametrize("pkg", ["mkckks", "mkbfv"])
def test_gpu_evaluator_keeps_the_references_signatures(pkg):
    funcs, methods, srcs = dropin(pkg)
    problems = check_evaluator(pkg, golden(), funcs, methods, srcs)
    assert not problems, "\n".join(problems)
    # without the tag the same names exist (alias of the reference's evaluator): patched tests build either way
    f_off, _, s_off = dropin(pkg, tagged=False)
    assert "NewGPUEvaluator" in f_off and any("type GPUEvaluator = Evaluator" in s for s in s_off.values())


def test_dropin_uses_only_wrappers_the_binding_has():
    """every method the drop-in calls on the engine context exists in package mkrlwegpu (shim/go/mkrlwegpu/*.go)"""
    have = set()
    for f in glob.glob(os.path.join(ROOT, "shim", "go", "mkrlwegpu", "*.go")):
        p = gen.parse_go_signatures(open(f).read())
        have |= set(p["methods"].get("Context", {})) | set(p["funcs"])
        for t in ("SwitchingKey", "Ciphertext"):
            have |= set(p["methods"].get(t, {}))
    used = set()
    for f in glob.glob(os.path.join(DROPIN, "*", "*_gpu.go")):
        src = open(f).read()
        used |= set(re.findall(r"\b(?:g|eval\.bfv|ks\.GPU\(\)|eval\.gpu\(\))\.([A-Z][A-Za-z]*)\(", src))
        used |= set(re.findall(r"\bmkrlwegpu\.([A-Z][A-Za-z]*)\(", src))
    missing = sorted(u for u in used if u not in have)
    assert not missing, "the drop-in calls %s, which package mkrlwegpu does not define" % missing
    assert {"Upload", "Download", "MulRelinRescale", "MulAndRelinHoisted", "RotateHoisted", "HoistedForm", "MulRelinBFV"} <= used
    # and the binding is a leaf: the in-package drop-in imports it, so it may import nothing of mk-lattigo
    for f in glob.glob(os.path.join(ROOT, "shim", "go", "mkrlwegpu", "*.go")):
        assert not re.search(r'^\s*"mk-lattigo/', open(f).read(), flags=re.M), "%s imports a package of the reference (import cycle for in-package tests)" % f


def test_patches_are_what_they_say():
    tags = open(os.path.join(ROOT, "shim", "go", "patches", "mkrlwe_build_tags.diff")).read()
    touched = re.findall(r"^\+\+\+ b/(\S+)", tags, flags=re.M)
    assert touched == ["mkrlwe/keyswitch.go", "mkrlwe/keyswitch_hoisted.go"] == json.load(open(GOLDEN))["files"]["mkrlwe"]
    added = [l[1:] for l in tags.splitlines() if l.startswith("+") and not l.startswith("+++")]
    assert set(added) == {"//go:build !mkhe_gpu", "// +build !mkhe_gpu", ""} and not [l for l in tags.splitlines() if l.startswith("-") and not l.startswith("---")]
    for pkg in ("mkckks", "mkbfv"):
        d = open(os.path.join(ROOT, "shim", "go", "patches", "%s_tests_gpu_evaluator.diff" % pkg)).read()
        plus = [l[1:].strip() for l in d.splitlines() if l.startswith("+") and not l.startswith("+++")]
        assert plus == ["evaluator *GPUEvaluator", "testContext.evaluator = NewGPUEvaluator(testContext.params)"]


def test_checker_reports_a_dropped_parameter_a_changed_type_and_a_missing_method():
    gold = golden()
    funcs, methods, srcs = dropin("mkrlwe")
    bad = copy.deepcopy(methods)
    bad["KeySwitcher"]["MulAndRelinHoisted"]["params"].pop()                       # ctOut dropped
    bad["KeySwitcher"]["Decompose"]["params"][1][1] = "*rlwe.PolyQP"                # a *ring.Poly turned into something else
    del bad["KeySwitcher"]["RotateHoisted"]
    text = "\n".join(check_keyswitcher(gold, funcs, bad))
    assert "MulAndRelinHoisted" in text and "Decompose" in text and "RotateHoisted (mkrlwe/keyswitch_hoisted.go): missing" in text
    f2, m2, s2 = dropin("mkckks")
    bad2 = copy.deepcopy(m2)
    bad2["GPUEvaluator"]["RescaleNew"]["results"].pop()                             # the error result dropped
    del bad2["GPUEvaluator"]["MulRelinNew"]
    text = "\n".join(check_evaluator("mkckks", gold, f2, bad2, s2))
    assert "RescaleNew" in text and "MulRelinNew: missing" in text


def test_dropin_declares_no_name_the_package_already_has():
    """Go refuses a second declaration of a top-level name in a package: nothing the drop-in files declare (functions, Type.Method, types, variables) may
    exist in the files of the reference package that stay compiled under the tag (`declared` in the golden table; the two replaced mkrlwe files are out)"""
    decl = json.load(open(GOLDEN))["declared"]
    for pkg in ("mkrlwe", "mkckks", "mkbfv"):
        for tagged in (True, False):
            _, _, srcs = dropin(pkg, tagged)
            mine = set()
            for s in srcs.values():
                mine |= gen.declared_names(s)
            clash = sorted(mine & set(decl[pkg]))
            assert not clash, "package %s already declares %s" % (pkg, clash)
    # and the two builds of a package do not declare a name twice between the drop-in files themselves
    for pkg in ("mkckks", "mkbfv"):
        for tagged in (True, False):
            _, _, srcs = dropin(pkg, tagged)
            seen = {}
            for f, s in srcs.items():
                for n in gen.declared_names(s):
                    assert n not in seen, "%s declared in %s and %s" % (n, seen[n], f)
                    seen[n] = f


def test_patches_apply_to_the_reference(tmp_path):
    """where the reference is mounted (the build container): `patch --dry-run` of every shipped patch against the files it names -- what install.sh does on a
    maintainer's checkout.  Skipped on the GPU box (no /root/reference there)."""
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(ref) or shutil.which("patch") is None:
        pytest.skip("no reference tree / no patch(1) here")
    for name in sorted(os.listdir(os.path.join(ROOT, "shim", "go", "patches"))):
        d = open(os.path.join(ROOT, "shim", "go", "patches", name)).read()
        work = tmp_path / name
        for rel in re.findall(r"^\+\+\+ b/(\S+)", d, flags=re.M):
            (work / os.path.dirname(rel)).mkdir(parents=True, exist_ok=True)
            shutil.copy(os.path.join(ref, rel), work / rel)
        r = subprocess.run(["patch", "-d", str(work), "-p1", "--forward", "--dry-run"], input=d, capture_output=True, text=True)
        assert r.returncode == 0, "%s does not apply:\n%s%s" % (name, r.stdout, r.stderr)
