#!/usr/bin/env python3
"""A few of the Go compiler's hard errors, checked without a Go toolchain (there is none in this image): the Go side of the shim has never been
compiled, and `imported and not used` / `declared and not used` / a wrong argument count are the mistakes an un-compiled Go file is most likely to hold.

    python3 tools/go_lint.py shim/go [1.13]     # prints the problems, exit 1 if any; the second argument is the module's `go` directive

Checks, per file (comments, strings and runes are removed by a small lexer first):
  * the file has a package clause; (), [], {} balance;
  * nothing newer than the module's `go` directive is used (the reference's go.mod says 1.13: no unsafe.Slice, no `any`, no generics, and every
    `//go:build` line has its `// +build` twin);
  * every import is used (`name.` occurs; `_` and `C` -- cgo -- are exempt: `C` must be used too when imported);
  * every local name introduced by `:=`, `var` or a `for ... := range` clause inside a function body occurs again in that body (an approximation of
    "declared and not used": a name that is only ever assigned passes here and fails in Go);
  * calls of the cgo binding's exported wrappers (package mkrlwegpu) from OTHER packages have an argument count some wrapper of that name accepts
    (calls are recognised by their receiver: `g.` after `g := ks.GPU()`, `ks.GPU().`, `eval.gpu().`, `eval.bfv.`, `mkrlwegpu.`); a call of a name the binding
    does not export through such a receiver is reported too;
  * calls from the drop-in INTO the reference's packages -- `NewCiphertext(...)`, `mkrlwe.NewSwitchingKey(...)`, `.GetRotationKey(...)` -- have an argument
    count the reference declares for that name (tests/golden/ref_go_signatures.json, `arities`: every function and method of mkrlwe / mkckks / mkbfv);
  * every selector of the drop-in that is not a call (`eval.ksw`, `rlk.Value`, `ct0.Scale`, `mkrlwegpu.RelinKeys`) names a field / type / method value that the
    reference (golden `fields`, `declared`), the shim itself or a short list of lattigo / standard-library names declares.
tests/test_go_lint_static.py runs it over shim/go and shows on doctored sources that each class of mistake is reported."""
import glob
import os
import re
import sys

IDENT = r"[A-Za-z_][A-Za-z0-9_]*"


def strip(src):
    """Go source -> same length text with comments, string / rune literals blanked (newlines kept)"""
    out = []
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        two = src[i:i + 2]
        if two == "//":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif two == "/*":
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append(re.sub(r"[^\n]", " ", src[i:j])); i = j
        elif c == "`":
            j = src.find("`", i + 1)
            j = n if j < 0 else j + 1
            out.append('""' + re.sub(r"[^\n]", " ", src[i + 2:j])); i = j
        elif c == '"' or c == "'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            j += 1
            out.append('""' + " " * (j - i - 2) if c == '"' else "0" + " " * (j - i - 1)); i = j
        else:
            out.append(c); i += 1
    return "".join(out)


def balance(text):
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    line = 1
    for ch in text:
        if ch == "\n":
            line += 1
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in ")]}":
            if not stack or stack[-1][0] != pairs[ch]:
                return "unbalanced %r at line %d" % (ch, line)
            stack.pop()
    if stack:
        return "unclosed %r opened at line %d" % stack[-1]
    return None


def imports(src_raw, text):
    """-> [(name, path)] ; needs the raw source for the paths (strings are blanked in text)"""
    out = []
    for m in re.finditer(r"^import\s*\(", text, flags=re.M):
        j = text.index(")", m.end())
        for ln in src_raw[m.end():j].split("\n"):
            mm = re.match(r'\s*(?:(%s|_|\.)\s+)?"([^"]+)"' % IDENT, ln)
            if mm:
                out.append((mm.group(1) or mm.group(2).split("/")[-1], mm.group(2)))
    for m in re.finditer(r'^import\s+(?:(%s|_|\.)\s+)?"' % IDENT, text, flags=re.M):
        mm = re.match(r'import\s+(?:(%s|_|\.)\s+)?"([^"]+)"' % IDENT, src_raw[m.start():])
        if mm:
            out.append((mm.group(1) or mm.group(2).split("/")[-1], mm.group(2)))
    return out


def match_close(text, i, open_ch="{", close_ch="}"):
    depth = 0
    while i < len(text):
        if text[i] == open_ch:
            depth += 1
        elif text[i] == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    return len(text)


def func_bodies(text):
    """-> [(name, header_text, body_text, line_of_func)] for every top-level func with a body"""
    out = []
    for m in re.finditer(r"^func\b", text, flags=re.M):
        i = m.end()
        # skip receiver, name, parameter list(s) up to the body's "{" at paren depth 0
        depth = 0
        j = i
        while j < len(text):
            ch = text[j]
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            elif ch == "{" and depth == 0:
                # `interface{}` / `struct{}` in a signature: an empty brace pair directly after the keyword
                before = text[i:j].rstrip()
                if before.endswith("interface") or before.endswith("struct"):
                    j = match_close(text, j)
                    continue
                break
            elif ch == "\n" and depth == 0 and text[j + 1:j + 2] not in (" ", "\t", ")"):
                j = -1
                break
            j += 1
        if j < 0 or j >= len(text):
            continue
        k = match_close(text, j)
        hdr = text[i:j]
        names = re.findall(r"(%s)\s*\(" % IDENT, re.sub(r"^\s*\([^()]*\)", "", hdr))        # receiver dropped: the first `name(` is the function's
        name = names[0] if names else "?"
        out.append((name, hdr, text[j:k], text.count("\n", 0, m.start()) + 1))
    return out


def declared_locals(body):
    """names introduced in a function body: `a, b := ...`, `var a, b T`, `for i, v := range` -- with the offset just past the declaration token"""
    out = []
    for m in re.finditer(r"((?:%s\s*,\s*)*%s)\s*:=" % (IDENT, IDENT), body):
        for nm in re.split(r"\s*,\s*", m.group(1).strip()):
            out.append((nm, m.start(1), m.end()))
    for m in re.finditer(r"\bvar\s+((?:%s\s*,\s*)*%s)\b" % (IDENT, IDENT), body):
        for nm in re.split(r"\s*,\s*", m.group(1).strip()):
            out.append((nm, m.start(1), m.end()))
    return out


def unused_locals(body):
    problems = []
    seen = set()
    for nm, s, e in declared_locals(body):
        if nm == "_" or (nm, s) in seen:
            continue
        seen.add((nm, s))
        # any other occurrence of the identifier in the body (not as a selector `.nm`, not as a struct-literal key `nm:`)
        uses = 0
        for u in re.finditer(r"(?<![A-Za-z0-9_.])%s\b" % re.escape(nm), body):
            if s <= u.start() < e and body[u.start():u.end()] == nm and u.start() < e and u.start() >= s:
                # inside the declaration's own name list
                continue
            uses += 1
        if uses == 0:
            problems.append("`%s` declared and not used (line +%d of the function)" % (nm, body.count("\n", 0, s)))
    return problems


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def wrapper_arities(binding_texts):
    """exported funcs / methods of the binding -> {name: set of (min_args, max_args or None)}"""
    ar = {}
    for text in binding_texts:
        for m in re.finditer(r"^func\s*(\([^)]*\)\s*)?(%s)\s*\(" % IDENT, text, flags=re.M):
            name = m.group(2)
            if not name[:1].isupper():
                continue
            i = m.end() - 1
            j = match_close(text, i, "(", ")")
            params = [p.strip() for p in split_args(text[i + 1:j - 1]) if p.strip()]
            variadic = bool(params) and "..." in params[-1]
            n = len(params)
            ar.setdefault(name, set()).add((n - 1 if variadic else n, None if variadic else n))
    return ar


# how the drop-in reaches the binding: `g := ks.GPU()` / `eval.gpu()` / `eval.bfv` (a *mkrlwegpu.Context) or the package itself
BINDING_RECEIVER = r"(?:\bg|GPU\(\)|gpu\(\)|\.bfv|\bmkrlwegpu)"


def check_calls(text, arities, ambiguous=()):
    problems = []
    for m in re.finditer(r"%s\.(%s)\s*\(" % (BINDING_RECEIVER, IDENT), text):
        name = m.group(1)
        if name in ambiguous:
            continue
        if name not in arities:
            if name[:1].isupper() and not re.match(r"[A-Z][A-Za-z0-9]*\s*\{", text[m.start(1):]):
                problems.append("line %d: %s is not an exported func / method of the binding" % (text.count("\n", 0, m.start()) + 1, name))
            continue
        i = m.end() - 1
        j = match_close(text, i, "(", ")")
        args = [a for a in split_args(text[i + 1:j - 1]) if a.strip()]
        n = len(args)
        spread = bool(args) and args[-1].rstrip().endswith("...")
        ok = any((lo <= n and (hi is None or n <= hi)) or (spread and hi is None and n == lo + 1) for lo, hi in arities[name])
        if not ok:
            problems.append("line %d: call of %s with %d argument(s); the binding takes %s" %
                            (text.count("\n", 0, m.start()) + 1, name, n, sorted("%d%s" % (lo, "+" if hi is None else "") for lo, hi in arities[name])))
    return problems


# language / library features newer than a go.mod `go` directive allows: (first version, pattern on the stripped text, what)
NEWER = [((1, 17), r"\bunsafe\.(Slice|Add)\s*\(", "unsafe.Slice / unsafe.Add"),
         ((1, 20), r"\bunsafe\.(String|StringData|SliceData)\s*\(", "unsafe.String / SliceData"),
         ((1, 18), r"^func\s*(\([^)]*\)\s*)?%s\s*\[" % IDENT, "type parameters"),
         ((1, 18), r"(?<![A-Za-z0-9_.])any\b(?!\s*(:?=|\())", "the predeclared `any`"),
         ((1, 21), r"\bruntime\.Pinner\b", "runtime.Pinner"),
         ((1, 21), r"(?<![A-Za-z0-9_.])(min|max|clear)\s*\(", "the builtins min / max / clear"),
         ((1, 19), r"\batomic\.(Int32|Int64|Uint32|Uint64|Bool|Pointer)\b", "the typed atomics")]


def version_problems(src, text, go):
    problems = []
    for since, pat, what in NEWER:
        if go < since:
            for m in re.finditer(pat, text, flags=re.M):
                problems.append("line %d: %s needs go %d.%d, the module is at go %d.%d" % (text.count("\n", 0, m.start()) + 1, what, since[0], since[1], go[0], go[1]))
    if go < (1, 17):
        # toolchains before 1.17 only read the old form: every //go:build line needs its `// +build` twin (same simple expression) right after it
        lines = src.split("\n")
        for i, l in enumerate(lines):
            m = re.match(r"^//go:build (.+)$", l)
            if m and not (i + 1 < len(lines) and lines[i + 1] == "// +build " + m.group(1)):
                problems.append("line %d: `//go:build %s` without the `// +build %s` line a go %d.%d module's toolchains may need" % (i + 1, m.group(1), m.group(1), go[0], go[1]))
    return problems


def check_ref_calls(text, pkg, ref, own=(), binding=()):
    """calls INTO the reference's packages from a drop-in file of package `pkg`: `Name(` (a function of the same package), `mkrlwe.Name(` (a function of
    that package) and `.Name(` (a method: the bare name over all receivers of mkrlwe / mkckks / mkbfv) against the parameter counts of the golden table
    (`arities`).  Names the drop-in declares itself (`own`) or the binding exports (`binding`: a call may be the binding's) are left alone."""
    problems = []
    def ok(n, spread, ranges):
        return any((lo <= n and (hi is None or n <= hi)) or (spread and hi is None and n == lo + 1) for lo, hi in ranges)
    def count(i):
        j = match_close(text, i, "(", ")")
        args = [x for x in split_args(text[i + 1:j - 1]) if x.strip()]
        return len(args), bool(args) and args[-1].rstrip().endswith("...")
    line = lambda pos: text.count("\n", 0, pos) + 1
    for m in re.finditer(r"(?<![A-Za-z0-9_.])(?:(mkrlwe|mkckks|mkbfv)\.)?(%s)\s*\(" % IDENT, text):
        q, name = m.group(1), m.group(2)
        if text[max(0, m.start() - 5):m.start()].strip().endswith("func"):
            continue                                             # a declaration, not a call
        table = ref.get(q or pkg, {}).get("funcs", {})
        if name in table and (q or name not in own):
            n, sp = count(m.end() - 1)
            if not ok(n, sp, table[name]):
                problems.append("line %d: %s%s called with %d argument(s); the reference declares %s" % (line(m.start()), (q + ".") if q else "", name, n, table[name]))
    methods = {}
    for p_ in ("mkrlwe", "mkckks", "mkbfv"):
        for k, v in ref.get(p_, {}).get("methods", {}).items():
            methods.setdefault(k, []).extend(v)
    for m in re.finditer(r"\.(%s)\s*\(" % IDENT, text):
        name = m.group(1)
        if name not in methods or name in own or name in binding or re.search(r"\b(mkrlwe|mkckks|mkbfv|mkrlwegpu)$", text[:m.start()]):
            continue
        n, sp = count(m.end() - 1)
        if not ok(n, sp, methods[name]):
            problems.append("line %d: method %s called with %d argument(s); the reference's methods of that name take %s" % (line(m.start()), name, n, methods[name]))
    return problems


# fields / types of lattigo v2.3.0 and the standard library the shim selects (the module is not vendored: this list is the whole of what is trusted unseen)
LATTIGO_FIELDS = {"Coeffs", "IsNTT", "Modulus", "MredParams", "BredParams", "NttPsi", "NttPsiInv", "N", "Q", "P", "Poly", "PolyQP", "Ring", "Mutex", "Pool", "Baseconverter",
                  "GaussianSampler", "Sampler", "Parameters", "KeySwitcher", "TestPN13QP218"}


def check_selectors(text, known):
    """every `.name` that is not a call must be a field / type / method value something declares: the reference's structs and top-level names (golden
    `fields`, `declared`), the shim's own declarations, or the short lattigo / standard-library list above -- a typo in a field name is a compile error"""
    problems = []
    for m in re.finditer(r"\.(%s)\b(?!\s*\()" % IDENT, text):
        name = m.group(1)
        if re.search(r"(?<![A-Za-z0-9_])[0-9]+$", text[max(0, m.start() - 24):m.start()]):
            continue                                             # a float literal
        if name not in known:
            problems.append("line %d: selector .%s names no field, type or method the reference, the shim or the lattigo list declares" % (text.count("\n", 0, m.start()) + 1, name))
    return problems


def lint_file(path, src=None, arities=None, ambiguous=(), go=None):
    src = open(path).read() if src is None else src
    text = strip(src)
    problems = version_problems(src, text, go) if go else []
    if not re.search(r"^package\s+%s\s*$" % IDENT, text, flags=re.M):
        problems.append("no package clause")
    b = balance(text)
    if b:
        problems.append(b)
        return problems
    body_text = re.sub(r"^import\s*\((?:[^)]*)\)", "", text, flags=re.M)
    body_text = re.sub(r'^import\s+[^\n]*', "", body_text, flags=re.M)
    for name, ipath in imports(src, text):
        if name in ("_", "."):
            continue
        if not re.search(r"(?<![A-Za-z0-9_.])%s\s*\." % re.escape(name), body_text):
            problems.append('"%s" imported and not used' % ipath)
    for name, hdr, body, line in func_bodies(text):
        for p in unused_locals(body):
            problems.append("func %s (line %d): %s" % (name, line, p))
    if arities:
        problems += check_calls(text, arities, ambiguous)
    return problems


def shim_declared(files):
    """names the shim's own files declare: struct fields, types, funcs / methods, package-level vars"""
    names = set()
    for f in files:
        t = strip(open(f).read())
        names |= set(re.findall(r"^func\s*(?:\([^)]*\)\s*)?(%s)\s*\(" % IDENT, t, flags=re.M))
        names |= set(re.findall(r"^type\s+(%s)" % IDENT, t, flags=re.M))
        for sm in re.finditer(r"^type\s+%s\s+struct\s*\{(.*?)^\}" % IDENT, t, flags=re.M | re.S):
            for ln in sm.group(1).split("\n"):
                fm = re.match(r"^\s*((?:%s\s*,\s*)*%s)\s+\S" % (IDENT, IDENT), ln)
                if fm:
                    names |= {x.strip() for x in fm.group(1).split(",")}
                else:
                    em = re.match(r"^\s*\*?(?:%s\.)?(%s)\s*$" % (IDENT, IDENT), ln)
                    if em:
                        names.add(em.group(1))
    return names


def lint_tree(root, go=None, ref=None, fields=None, declared=None):
    files = sorted(glob.glob(os.path.join(root, "**", "*.go"), recursive=True))
    binding = [f for f in files if os.sep + "mkrlwegpu" + os.sep in f]
    arities = wrapper_arities([strip(open(f).read()) for f in binding])
    known = set(LATTIGO_FIELDS) | shim_declared(files)
    for table in (fields or {}), (declared or {}):
        for v in table.values():
            known |= {x.split(".")[-1] for x in v}
    out = {}
    for f in files:
        inside = f in binding
        if ref and os.sep + "dropin" + os.sep in f:
            pkg = os.path.basename(os.path.dirname(f))
            own = set()
            for g in glob.glob(os.path.join(os.path.dirname(f), "*.go")):
                own |= set(re.findall(r"^func\s*(?:\([^)]*\)\s*)?(%s)\s*\(" % IDENT, strip(open(g).read()), flags=re.M))
            pr = check_ref_calls(strip(open(f).read()), pkg, ref, own=own, binding=set(arities))
            if fields is not None:
                pr += check_selectors(strip(open(f).read()), known)
            if pr:
                out.setdefault(f, []).extend(pr)
        # names the reference's packages (and Go's own) use too: a call `.Name(` in a drop-in file may not be the binding's
        p = lint_file(f, arities=None if inside else arities, go=go)
        if p:
            out.setdefault(f, []).extend(p)
    return out


if __name__ == "__main__":
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_go_signatures.json")
    import json
    res = lint_tree(sys.argv[1] if len(sys.argv) > 1 else "shim/go", go=tuple(int(x) for x in sys.argv[2].split(".")) if len(sys.argv) > 2 else None,
                    **({k2: json.load(open(gold)).get(k1) for k1, k2 in (("arities", "ref"), ("fields", "fields"), ("declared", "declared"))} if os.path.exists(gold) else {}))
    for f, ps in res.items():
        for p in ps:
            print("%s: %s" % (f, p))
    sys.exit(1 if res else 0)
