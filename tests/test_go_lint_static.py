"""The Go side of the shim (shim/go/**, never compiled: no Go toolchain in this image) against a few of the Go compiler's hard errors that need no
toolchain to find -- unused imports, locals that are declared and never mentioned again, unbalanced brackets, and calls of the cgo binding from the
drop-in with an argument count no wrapper of that name takes (tools/go_lint.py).  The second half shows on doctored copies of the shipped files that
each class of mistake IS reported: a lint that cannot fail proves nothing."""
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "go")

_spec = importlib.util.spec_from_file_location("go_lint", os.path.join(ROOT, "tools", "go_lint.py"))
lint = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(lint)

KS = os.path.join(SHIM, "dropin", "mkrlwe", "keyswitch_gpu.go")
BIND = os.path.join(SHIM, "mkrlwegpu", "mkrlwegpu.go")


def arities():
    import glob
    return lint.wrapper_arities([lint.strip(open(f).read()) for f in sorted(glob.glob(os.path.join(SHIM, "mkrlwegpu", "*.go")))])


def go_directive():
    """the `go` line of the reference's go.mod (tests/golden/ref_go_signatures.json, written by tools/gen_ref_go_signatures.py)"""
    return tuple(json.load(open(os.path.join(ROOT, "tests", "golden", "ref_go_signatures.json")))["go_directive"])


def ref_arities():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "ref_go_signatures.json")))["arities"]


def test_shim_tree_is_clean():
    assert go_directive() == (1, 13)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_go_signatures.json")))
    res = lint.lint_tree(SHIM, go=go_directive(), ref=gold["arities"], fields=gold["fields"], declared=gold["declared"])
    assert not res, "\n".join("%s: %s" % (os.path.relpath(f, ROOT), p) for f, ps in res.items() for p in ps)


def test_the_lint_sees_every_function_and_every_binding_call():
    """the checks only mean something if the parser finds the code: every `func` of the drop-in has a body here, and the drop-in's engine calls are seen"""
    src = open(KS).read()
    text = lint.strip(src)
    bodies = lint.func_bodies(text)
    assert len(bodies) == len(re.findall(r"^func\b", text, flags=re.M)) and len(bodies) >= 20
    names = {n for n, _, _, _ in bodies}
    assert {"NewKeySwitcher", "GPU", "Decompose", "MulAndRelinHoisted", "RotateHoisted", "Conjugate", "ExternalProduct"} <= names
    calls = re.findall(r"%s\.(%s)\s*\(" % (lint.BINDING_RECEIVER, lint.IDENT), text)
    assert {"MulAndRelinHoisted", "RotateHoisted", "Conjugate", "UploadSwitchingKey", "NewCiphertext"} <= set(calls)
    ar = arities()
    assert len(ar) >= 45 and ar["MulAndRelinHoisted"] == {(7, 7)} and ar["SortedIDs"] == {(1, 1)}


def test_unused_import_is_reported():
    src = open(KS).read().replace('import (\n', 'import (\n\t"strings"\n', 1)
    assert any('"strings" imported and not used' in p for p in lint.lint_file(KS, src))
    # ... and cgo's pseudo-package counts as an import that must be used
    bsrc = re.sub(r"\bC\.", "X.", open(BIND).read())
    assert any('"C" imported and not used' in p for p in lint.lint_file(BIND, bsrc))


def test_unused_local_is_reported():
    src = open(KS).read()
    doctored = src.replace("\tg := ks.GPU()\n", "\tg := ks.GPU()\n\tleftover := 3\n", 1)
    assert doctored != src
    probs = lint.lint_file(KS, doctored)
    assert any("`leftover` declared and not used" in p for p in probs), probs
    doctored = src.replace("\tg := ks.GPU()\n", "\tg := ks.GPU()\n\tvar spare []uint64\n", 1)
    assert any("`spare` declared and not used" in p for p in lint.lint_file(KS, doctored))


def test_wrong_argument_count_and_unknown_wrapper_are_reported():
    src = open(KS).read()
    text = lint.strip(src)
    m = re.search(r"\bg\.MulAndRelinHoisted\(", text)
    assert m
    end = lint.match_close(text, m.end() - 1, "(", ")")
    args = lint.split_args(src[m.end():end - 1])
    assert len(args) == 7
    doctored = src[:m.end()] + ",".join(args[:-1]) + src[end - 1:]
    probs = lint.lint_file(KS, doctored, arities=arities())
    assert any("call of MulAndRelinHoisted with 6 argument(s)" in p for p in probs), probs
    doctored = src.replace("g.MulAndRelinHoisted(", "g.MulAndRelinFused(", 1)
    probs = lint.lint_file(KS, doctored, arities=arities())
    assert any("MulAndRelinFused is not an exported func / method of the binding" in p for p in probs), probs


def test_unbalanced_brackets_and_missing_package_clause_are_reported():
    src = open(KS).read()
    i = src.rindex("}")
    assert any("unclosed" in p or "unbalanced" in p for p in lint.lint_file(KS, src[:i] + src[i + 1:]))
    assert any("no package clause" in p for p in lint.lint_file(KS, re.sub(r"^package mkrlwe$", "", src, count=1, flags=re.M)))


def test_literals_and_comments_do_not_confuse_the_lexer():
    src = 'package x\n\nimport "fmt"\n\n// a } in a comment, an unused := in a comment: y := 1\nfunc f() {\n\ts := "} not a brace \\" { "\n\tr := \'}\'\n\traw := `{{{`\n\tfmt.Println(s, r, raw)\n}\n'
    assert lint.lint_file("x.go", src) == []


def test_features_newer_than_the_modules_go_directive_are_reported():
    """the reference's go.mod says `go 1.13`: unsafe.Slice (1.17), `any` (1.18) and a //go:build line without its // +build twin do not compile / are not
    honoured there -- the first two were in the shim until round 5"""
    src = open(os.path.join(SHIM, "mkrlwegpu", "keyswitchgpu.go")).read()
    go = go_directive()
    doctored = src.replace("return (*[1 << 28]*C.mkhe_swk)(unsafe.Pointer(arr))[:n:n]", "return unsafe.Slice(arr, n)", 1)
    assert doctored != src and any("unsafe.Slice" in p and "needs go 1.17" in p for p in lint.lint_file("k.go", doctored, go=go))
    assert not any("needs go" in p for p in lint.lint_file("k.go", doctored, go=(1, 21)))
    doctored = src.replace("// +build mkhe_gpu\n", "", 1)
    assert doctored != src and any("without the `// +build mkhe_gpu` line" in p for p in lint.lint_file("k.go", doctored, go=go))
    doctored = src.replace("func b2i(b bool) C.int {", "func b2i(b bool, _ any) C.int {", 1)
    assert doctored != src and any("`any`" in p for p in lint.lint_file("k.go", doctored, go=go))


def test_calls_into_the_reference_with_a_wrong_argument_count_are_reported():
    """the drop-in calls the reference's own constructors and accessors (NewCiphertext has four parameters in mkckks and two in mkbfv; GetRotationKey two);
    the golden table carries every function's and method's parameter count, the lint holds the calls against it"""
    ref = ref_arities()
    assert ref["mkckks"]["funcs"]["NewCiphertext"] == [[4, 4]] and ref["mkbfv"]["funcs"]["NewCiphertext"] == [[2, 2]]
    assert ref["mkrlwe"]["methods"]["GetRotationKey"] == [[2, 2]] and ref["mkrlwe"]["funcs"]["NewSwitchingKey"] == [[1, 1]]
    ev = os.path.join(SHIM, "dropin", "mkckks", "evaluator_gpu.go")
    text = lint.strip(open(ev).read())
    bind = set(arities())
    assert lint.check_ref_calls(text, "mkckks", ref, binding=bind) == []
    # the checks see the calls they are about
    assert len(re.findall(r"(?<![A-Za-z0-9_.])NewCiphertext\s*\(", text)) >= 5 and "GetRotationKey(" in text and "mkrlwe.NewHoistedCiphertext(" in text
    for old, new, what in (("NewCiphertext(eval.params, ct0.IDSet(), ct0.Level(), ct0.Scale)", "NewCiphertext(eval.params, ct0.IDSet(), ct0.Level())", "NewCiphertext called with 3"),
                           ("rkSet.GetRotationKey(id, uint(rotidx))", "rkSet.GetRotationKey(id)", "method GetRotationKey called with 1"),
                           ("mkrlwe.NewHoistedCiphertext()", "mkrlwe.NewHoistedCiphertext(eval.params)", "mkrlwe.NewHoistedCiphertext called with 1"),
                           ("op0.IDSet().Union(op1.IDSet())", "op0.IDSet().Union()", "method Union called with 0")):
        src = open(ev).read()
        assert old in src, old
        probs = lint.check_ref_calls(lint.strip(src.replace(old, new, 1)), "mkckks", ref, binding=bind)
        assert any(what in p for p in probs), (what, probs)


def test_a_misspelt_field_is_reported():
    """`eval.ksw`, `eval.params`, `rlk.Value`, `ct0.Scale` ... are fields of the reference's structs (golden `fields`: names only); a selector that names nothing
    the reference, the shim or the short lattigo list declares is a compile error waiting for the first machine with Go"""
    import glob
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_go_signatures.json")))
    assert {"ksw", "params", "Scale"} <= set(gold["fields"]["mkckks"]) and {"CRS", "Value", "Decomposer"} <= set(gold["fields"]["mkrlwe"])
    known = set(lint.LATTIGO_FIELDS) | lint.shim_declared(sorted(glob.glob(os.path.join(SHIM, "**", "*.go"), recursive=True)))
    for table in (gold["fields"], gold["declared"]):
        for v in table.values():
            known |= {x.split(".")[-1] for x in v}
    ev = os.path.join(SHIM, "dropin", "mkckks", "evaluator_gpu.go")
    src = open(ev).read()
    assert lint.check_selectors(lint.strip(src), known) == []
    for old, new in (("eval.ksw.HostMirror = false", "eval.kws.HostMirror = false"), ("eval.params.CRS[-1]", "eval.params.Crs[-1]"), ("ct0.Scale == 0", "ct0.Scal == 0")):
        assert old in src
        probs = lint.check_selectors(lint.strip(src.replace(old, new, 1)), known)
        assert len(probs) == 1 and "selector ." in probs[0], probs
