#!/usr/bin/env python3
"""Writes tests/golden/ref_go_signatures.json: the exported function / method SIGNATURES of the reference files whose API the Go drop-in
(shim/go/dropin) must reproduce -- a table of names and parameter / result types, not source text.  Run in the build container, where the
reference is mounted read-only at /root/reference:

    python3 tools/gen_ref_go_signatures.py [/root/reference] > tests/golden/ref_go_signatures.json

tests/test_go_dropin_static.py parses shim/go/dropin/**/*.go with the same parser (parse_go_signatures below) and compares."""
import json
import os
import re
import sys

FILES = {
    "mkrlwe": ["mkrlwe/keyswitch.go", "mkrlwe/keyswitch_hoisted.go"],
    "mkckks": ["mkckks/evaluator.go"],
    "mkbfv": ["mkbfv/evaluator.go", "mkbfv/keyswitch.go", "mkbfv/keyswitch_hoisted.go"],
}


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _split_top(s):
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


def _balanced(src, i):
    """src[i] == '(' -> index just past the matching ')'"""
    depth = 0
    while True:
        depth += {"(": 1, ")": -1}.get(src[i], 0)
        i += 1
        if depth == 0:
            return i


def _fields(text):
    """Go parameter / result list -> [[name, type], ...] with grouped names expanded (`a, b int` -> a int, b int); unnamed lists -> ["", type]"""
    parts = [" ".join(p.split()) for p in _split_top(text)]
    if not parts:
        return []
    # named iff some part is `name type...`: two or more words whose first is an identifier and the rest does not start with `.` (pkg.Type is ONE word)
    def split_named(p):
        m = re.match(r"^([A-Za-z_][A-Za-z0-9_]*)\s+(.+)$", p)
        return (m.group(1), m.group(2)) if m and (m.group(2).startswith("...") or not m.group(2).startswith(".")) and m.group(1) not in ("func", "map", "chan", "struct", "interface") else None
    named = any(split_named(p) for p in parts)
    out, pending = [], []
    for p in parts:
        sn = split_named(p) if named else None
        if sn:
            for n in pending:
                out.append([n, sn[1]])
            pending = []
            out.append([sn[0], sn[1]])
        elif named:
            pending.append(p)
        else:
            out.append(["", p])
    if pending:
        raise ValueError("unparsed names %r in %r" % (pending, text))
    return out


def parse_go_signatures(src):
    """-> {"funcs": {name: sig}, "methods": {Type: {name: sig}}} with sig = {"params": [[name, type]...], "results": [[name, type]...]}"""
    src = _strip_comments(src)
    funcs, methods = {}, {}
    for m in re.finditer(r"^func\s*", src, flags=re.M):
        i = m.end()
        recv = None
        if src[i] == "(":
            j = _balanced(src, i)
            r = _fields(src[i + 1:j - 1])
            recv = r[0][1].lstrip("*").strip() if r else None
            i = j
        mm = re.match(r"\s*([A-Za-z_][A-Za-z0-9_]*)\s*", src[i:])
        if not mm:
            continue
        name = mm.group(1)
        i += mm.end()
        if src[i] != "(":
            continue
        j = _balanced(src, i)
        params = _fields(src[i + 1:j - 1])
        rest = src[j:]
        k = rest.index("{")
        res_txt = rest[:k].strip()
        if res_txt.startswith("("):
            results = _fields(res_txt[1:_balanced(res_txt, 0) - 1])
        else:
            results = [["", " ".join(res_txt.split())]] if res_txt else []
        sig = {"params": params, "results": results}
        if recv:
            methods.setdefault(recv, {})[name] = sig
        else:
            funcs[name] = sig
    return {"funcs": funcs, "methods": methods}


def exported(d):
    return {k: v for k, v in d.items() if k[:1].isupper()}


# files the build tag excludes when the drop-in is active (shim/go/patches/mkrlwe_build_tags.diff): their names are free again
EXCLUDED = {"mkrlwe": ["mkrlwe/keyswitch.go", "mkrlwe/keyswitch_hoisted.go"]}


def declared_names(src):
    """top-level identifiers a Go file declares: funcs, Type.Method, types, package-level vars / consts (single and grouped)"""
    src = _strip_comments(src)
    p = parse_go_signatures(src)
    names = set(p["funcs"])
    for t, ms in p["methods"].items():
        names |= {"%s.%s" % (t, m) for m in ms}
    names |= set(re.findall(r"^type\s+([A-Za-z_][A-Za-z0-9_]*)", src, flags=re.M))
    names |= set(re.findall(r"^(?:var|const)\s+([A-Za-z_][A-Za-z0-9_]*)", src, flags=re.M))
    for m in re.finditer(r"^(?:var|const)\s*\((.*?)^\)", src, flags=re.M | re.S):
        names |= set(re.findall(r"^\s*([A-Za-z_][A-Za-z0-9_]*)", m.group(1), flags=re.M))
    return names


def build(ref_root):
    table = {}
    for pkg, files in FILES.items():
        funcs, methods = {}, {}
        for f in files:
            p = parse_go_signatures(open(os.path.join(ref_root, f)).read())
            for k, v in exported(p["funcs"]).items():
                funcs[k] = dict(v, file=f)
            for t, ms in p["methods"].items():
                for k, v in exported(ms).items():
                    methods.setdefault(t, {})[k] = dict(v, file=f)
        table[pkg] = {"funcs": funcs, "methods": methods}
    declared = {}
    for pkg in FILES:
        names = set()
        for fn in sorted(os.listdir(os.path.join(ref_root, pkg))):
            rel = "%s/%s" % (pkg, fn)
            if fn.endswith(".go") and not fn.endswith("_test.go") and rel not in EXCLUDED.get(pkg, []):
                names |= declared_names(open(os.path.join(ref_root, rel)).read())
        declared[pkg] = sorted(names)
    # `declared`: every top-level name of the package's non-test files that stay compiled under the tag -- a drop-in file may not declare one again
    # `arities`: per package, the parameter counts of EVERY function and method of its non-test files (functions by name, methods by bare name over all
    # receivers; [min, max], max null for a variadic one) -- what tools/go_lint.py holds the drop-in's calls INTO the reference's packages against
    arities = {}
    for pkg in FILES:
        fn, me = {}, {}
        for name in sorted(os.listdir(os.path.join(ref_root, pkg))):
            if not name.endswith(".go") or name.endswith("_test.go"):
                continue
            p = parse_go_signatures(open(os.path.join(ref_root, pkg, name)).read())
            def rng(sig):
                n = len(sig["params"]); var = n > 0 and sig["params"][-1][1].startswith("...")
                return [n - 1 if var else n, None if var else n]
            for k, v in p["funcs"].items():
                if rng(v) not in fn.setdefault(k, []): fn[k].append(rng(v))
            for t, ms in p["methods"].items():
                for k, v in ms.items():
                    if rng(v) not in me.setdefault(k, []): me[k].append(rng(v))
        arities[pkg] = {"funcs": fn, "methods": me}
    # `fields`: per package, the field names of its struct types (embedded types by their type name) -- names only; go_lint holds the drop-in's selectors against them
    fields = {}
    for pkg in FILES:
        names = set()
        for name in sorted(os.listdir(os.path.join(ref_root, pkg))):
            if not name.endswith(".go") or name.endswith("_test.go"):
                continue
            src = _strip_comments(open(os.path.join(ref_root, pkg, name)).read())
            for sm in re.finditer(r"^type\s+[A-Za-z_][A-Za-z0-9_]*\s+struct\s*\{(.*?)^\}", src, flags=re.M | re.S):
                for ln in sm.group(1).split("\n"):
                    ln = ln.strip()
                    if not ln:
                        continue
                    fm = re.match(r"^((?:[A-Za-z_][A-Za-z0-9_]*\s*,\s*)*[A-Za-z_][A-Za-z0-9_]*)\s+\S", ln)
                    if fm:
                        names |= {x.strip() for x in fm.group(1).split(",")}
                    else:
                        em = re.match(r"^\*?(?:[A-Za-z_][A-Za-z0-9_]*\.)?([A-Za-z_][A-Za-z0-9_]*)$", ln)
                        if em:
                            names.add(em.group(1))
        fields[pkg] = sorted(names)
    m = re.search(r"^go\s+(\d+)\.(\d+)", open(os.path.join(ref_root, "go.mod")).read(), flags=re.M)
    # `go_directive`: the language version the module compiles at (go.mod) -- the shim may not use anything newer (tools/go_lint.py)
    return {"reference": "SNUCP/MKHE-KKLSS", "go_directive": [int(m.group(1)), int(m.group(2))], "arities": arities, "fields": fields, "files": FILES, "excluded_under_tag": EXCLUDED, "packages": table, "declared": declared}


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    json.dump(build(root), sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")
