"""The Go side of the boundary (shim/go/**, un-built here: no Go toolchain) against include/mkhe.h: every `C.mkhe_*(...)` call must name an exported
function, pass the right number of arguments, and pass a pointer where the header takes a pointer and a C scalar where it takes a scalar -- with the
pointed-to / scalar C type where the Go expression spells it (casts such as (*C.uint64_t)(...), C.int(...)).  A changed header signature fails here.
Also: every KeySwitcher / Evaluator method of SURVEY.md 8(b)'s table has a Go wrapper."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mkhe.h")
GO_FILES = sorted(glob.glob(os.path.join(ROOT, "shim", "go", "**", "*.go"), recursive=True))


def strip_c_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def parse_header():
    """name -> (return type, [(base type, pointer depth)])"""
    txt = strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"\b((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*)*)\s*\b(mkhe_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                depth = a.count("*")
                base = re.sub(r"\bconst\b|\*", " ", a).split()
                # drop the parameter name (last token) unless the declaration is unnamed
                base = base[:-1] if len(base) > 1 else base
                params.append((" ".join(base), depth))
        protos[name] = (ret.strip(), params)
    return protos


def split_top_level(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def go_calls(path):
    """yields (function name, [argument expressions], enclosing function source, line number)"""
    src = open(path).read()
    src_nc = re.sub(r"//[^\n]*", lambda m: " " * len(m.group(0)), src)          # keep offsets
    funcs = [(m.start(), m.group(0)) for m in re.finditer(r"^func [^\n]*\{", src_nc, flags=re.M)]
    for m in re.finditer(r"\bC\.(mkhe_[a-z0-9_]+)\(", src_nc):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src_nc[i], 0)
            i += 1
        args = split_top_level(" ".join(src_nc[m.end():i - 1].split()))
        start = max([s for s, _ in funcs if s <= m.start()], default=0)
        nxt = min([s for s, _ in funcs if s > m.start()], default=len(src_nc))
        yield m.group(1), args, src_nc[start:nxt], src_nc.count("\n", 0, m.start()) + 1


SCALAR_CAST = re.compile(r"^C\.(int|uint64_t|size_t|int32_t|long|double|uint)\(")
PTR_CAST = re.compile(r"^\((\*+)C\.([A-Za-z0-9_]+)\)\(")


def classify(expr, fn_src):
    """-> ('scalar', ctype or None) | ('pointer', (depth, ctype) or None) | None (undetermined)"""
    m = SCALAR_CAST.match(expr)
    if m:
        return "scalar", m.group(1)
    m = PTR_CAST.match(expr)
    if m:
        return "pointer", (len(m.group(1)), m.group(2))
    if expr == "nil" or expr.startswith("&"):
        return "pointer", None
    if re.fullmatch(r"[A-Za-z_][A-Za-z0-9_.]*\.(h|c|d|dev)", expr):     # handle fields: *C.mkhe_swk / *C.mkhe_ct / *C.mkhe_ctx / unsafe.Pointer (device buffers)
        return "pointer", None
    if re.fullmatch(r"b2i\(.*\)", expr):
        return "scalar", "int"
    if re.fullmatch(r"[A-Za-z_][A-Za-z0-9_]*", expr):
        name = re.escape(expr)
        # declared in the enclosing function (parameter list or body) with a recognisable type or initialiser
        if re.search(r"\b%s\b(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*)*\s+(?:unsafe\.Pointer|\*+C\.[A-Za-z0-9_]+)" % name, fn_src):
            return "pointer", None
        # handle arrays built by the shim's helpers: swkList / swkArray -> **C.mkhe_swk, ctArray -> **C.mkhe_ct (batchgpu.go); in a multiple
        # assignment every name on the left takes the helper of the statement (the shim only ever mixes one helper per statement)
        for helpers, ctype in ((r"(?:swkList|swkArray)", "mkhe_swk"), (r"ctArray", "mkhe_ct")):
            if re.search(r"(?:\b%s\b|\b[A-Za-z_0-9]+\s*,\s*%s\b|\b%s\s*,\s*[A-Za-z_0-9]+(?:\s*,\s*[A-Za-z_0-9]+)*|\b[A-Za-z_0-9]+\s*,\s*[A-Za-z_0-9]+\s*,\s*%s\b|\b[A-Za-z_0-9]+\s*,\s*%s\s*,\s*[A-Za-z_0-9]+)\s*:?=\s*%s\(" % (name, name, name, name, name, helpers), fn_src):
                return "pointer", (2, ctype)
            if re.search(r"\b%s\b[^\n]*:?=\s*[^\n]*%s\(" % (name, helpers), fn_src):
                return "pointer", (2, ctype)
        m2 = re.search(r"\b%s\s*:?=\s*\((\*+)C\.([A-Za-z0-9_]+)\)\(" % name, fn_src)
        if m2:
            return "pointer", (len(m2.group(1)), m2.group(2))
        if re.search(r"var\s+[^\n]*\b%s\b[^\n]*\*+C\." % name, fn_src):
            return "pointer", None
        if re.search(r"\b%s\s*:?=\s*C\.(int|uint64_t|size_t|int32_t)\(" % name, fn_src):
            return "scalar", None
    return None


def test_header_parses_every_export():
    protos = parse_header()
    assert len(protos) >= 89, len(protos)
    assert protos["mkhe_rotate"][1] == [("mkhe_ctx", 1), ("uint64_t", 0), ("mkhe_ct", 1), ("mkhe_swk", 2), ("mkhe_swk", 2), ("mkhe_swk", 1), ("mkhe_ct", 1)]
    assert protos["mkhe_ctx_sync"][1] == [("mkhe_ctx", 1)]


def check_calls(protos):
    checked, problems = 0, []
    for path in GO_FILES:
        rel = os.path.relpath(path, ROOT)
        for name, args, fn_src, line in go_calls(path):
            where = "%s:%d C.%s" % (rel, line, name)
            if name not in protos:
                problems.append("%s: not declared in include/mkhe.h" % where); continue
            params = protos[name][1]
            if len(args) != len(params):
                problems.append("%s: %d arguments, the header declares %d" % (where, len(args), len(params))); continue
            for k, (expr, (ctype, depth)) in enumerate(zip(args, params)):
                got = classify(expr, fn_src)
                if got is None:
                    problems.append("%s: argument %d `%s`: cannot tell pointer from scalar (use a C.<type>(...) / (*C.<type>)(...) form or a handle field)" % (where, k, expr))
                    continue
                kind, detail = got
                if (depth > 0) != (kind == "pointer"):
                    problems.append("%s: argument %d `%s` is a %s, the header takes %s%s" % (where, k, expr, kind, ctype, "*" * depth)); continue
                if kind == "scalar" and detail and detail != ctype:
                    problems.append("%s: argument %d `%s`: C.%s for a parameter of type %s" % (where, k, expr, detail, ctype))
                if kind == "pointer" and detail and not (ctype == "void" and depth == 1):
                    d, t = detail
                    if d != depth or t != ctype:
                        problems.append("%s: argument %d `%s`: %sC.%s for a parameter of type %s%s" % (where, k, expr, "*" * d, t, ctype, "*" * depth))
            checked += 1
    return checked, problems


def test_every_cgo_call_matches_the_header():
    assert GO_FILES, "shim/go/**/*.go not found"
    checked, problems = check_calls(parse_header())
    assert not problems, "\n".join(problems)
    assert checked >= 45, checked


def test_checker_detects_a_changed_signature():
    """the guard itself: a dropped parameter, a scalar turned pointer and a renamed export are all reported"""
    protos = parse_header()
    ret, params = protos["mkhe_rotate"]
    protos["mkhe_rotate"] = (ret, params[:-1])
    ret, params = protos["mkhe_rescale"]
    protos["mkhe_rescale"] = (ret, [params[0], params[1], ("int", 1), params[3]])
    del protos["mkhe_conjugate"]
    _, problems = check_calls(protos)
    text = "\n".join(problems)
    assert "C.mkhe_rotate: 7 arguments, the header declares 6" in text
    assert "C.mkhe_rescale: argument 2" in text and "is a scalar" in text
    assert "C.mkhe_conjugate: not declared" in text


def test_keyswitcher_method_set_has_go_wrappers():
    """one wrapper per reference method of the path (name of the Go method in package mkrlwegpu, the entry point it must reach)"""
    want = {
        "Decompose": "mkhe_decompose",                                  # mkrlwe/keyswitch.go:49
        "ExternalProduct": "mkhe_external_product",                     # :79
        "ExternalProductHoisted": "mkhe_external_product_hoisted",      # keyswitch_hoisted.go:10
        "MulAndRelin": "mkhe_mul_and_relin",                            # keyswitch.go:122
        "MulAndRelinHoisted": "mkhe_mul_and_relin",                     # keyswitch_hoisted.go:44
        "Rotate": "RotateHoisted",                                      # keyswitch.go:234 (delegates)
        "RotateHoisted": "mkhe_rotate",                                 # keyswitch_hoisted.go:183
        "Conjugate": "mkhe_conjugate",                                  # keyswitch.go:302
        "HoistedForm": "mkhe_hoisted_form",                             # mkckks/evaluator.go:543
        "Rescale": "mkhe_rescale",                                      # mkckks/evaluator.go:359
        "MulRelinRescale": "mkhe_mul_relin_rescale",                    # mkckks/evaluator.go:558-581
        "MultByConst": "mkhe_ct_mul_const",                             # mkckks/evaluator.go:117
        "MulPtxt": "mkhe_ct_mul_ptxt",                                  # mkckks/evaluator.go:465
        "Add": "mkhe_ct_add", "Sub": "mkhe_ct_sub",                     # mkckks/evaluator.go:316-356
        "MulRelinBFV": "mkhe_bfv_mul_relin",                            # mkbfv/evaluator.go:118
        "MulRelinBFVUnhoisted": "mkhe_bfv_mul_relin_unhoisted",         # mkbfv/evaluator.go:94
        "ExternalProductBFV": "mkhe_bfv_external_product",              # mkbfv/keyswitch.go:83
        "ExternalProductBFVHoisted": "mkhe_bfv_external_product_hoisted",   # mkbfv/keyswitch_hoisted.go:7
    }
    src = "\n".join(open(p).read() for p in GO_FILES if os.sep + "mkrlwegpu" + os.sep in p)
    for method, needle in want.items():
        m = re.search(r"^func \(ctx \*Context\) %s\(.*?^\}" % method, src, flags=re.M | re.S)
        assert m, "no Go wrapper for %s" % method
        assert needle in m.group(0), "%s does not reach %s" % (method, needle)
