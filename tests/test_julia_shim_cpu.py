"""Static check of julia/BEnlsipHIP.jl against include/benlsip_hip.h.

No Julia toolchain exists in this pipeline, so the shim cannot be executed; what CAN be checked mechanically is the part whose
failure mode is silent memory corruption: every `ccall((:bh_x, libbh), Ret, (types...), args...)` must name an export the
header declares, with the same arity, the same return type, and per argument the same class — Int32 / Int64 / Float64 by
value, or a pointer whose element type matches (`Ptr{Cvoid}` / `void*` match any pointer: handles and device addresses)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_top(s):
    """Split on commas that are not nested inside (), {} or []."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(text, i):
    """text[i] == '(' -> index just past its matching ')', skipping string literals."""
    depth, j, in_str = 0, i, False
    while j < len(text):
        ch = text[j]
        if in_str:
            if ch == "\\":
                j += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
        elif ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return j + 1
        j += 1
    raise ValueError("unbalanced parenthesis at %d" % i)


def julia_ccalls():
    text = open(os.path.join(ROOT, "julia", "BEnlsipHIP.jl")).read()
    text = "\n".join(ln if not ln.lstrip().startswith("#") else "" for ln in text.splitlines())
    text = re.sub(r"#[^\n\"]*$", "", text, flags=re.M)          # trailing comments (none of them contains a quote)
    calls = []
    for m in re.finditer(r"\bccall\(", text):
        end = _balanced(text, m.end() - 1)
        parts = _split_top(text[m.end():end - 1])
        sym = re.fullmatch(r"\(\s*:(\w+)\s*,\s*libbh\s*\)", parts[0])
        assert sym, "ccall with an unexpected target: %s" % parts[0]
        types = parts[2]
        assert types.startswith("(") and types.endswith(")"), types
        tlist = _split_top(types[1:-1])
        calls.append(dict(name=sym.group(1), ret=parts[1], types=tlist, nargs=len(parts) - 3, line=text.count("\n", 0, m.start()) + 1))
    return calls


def header_prototypes():
    text = open(os.path.join(ROOT, "include", "benlsip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int32_t)\s+(bh_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = dict(ret="cstring" if "char" in ret else "i32", params=[c_class(p) for p in params], raw=params)
    return protos


def c_class(param):
    p = param.replace("const ", "").strip()
    stars = p.count("*")
    base = re.match(r"(unsigned\s+long\s+long|\w+)", p).group(1)
    if stars == 0:
        return {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64"}[base]
    elem = {"double": "f64", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "void": "any", "char": "char",
            "bh_hess": "any", "bh_proj": "any", "bh_stats_t": "any"}[base]
    return "ptr:" + ("ptr" if stars == 2 else elem)


def julia_class(t):
    t = t.strip()
    by_value = {"Int32": "i32", "Int64": "i64", "UInt64": "u64", "Float64": "f64"}
    if t in by_value:
        return by_value[t]
    if t == "Cstring":
        return "ptr:char"
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.+)\}", t)
    assert m, "unrecognised ccall argument type %r" % t
    inner = m.group(1)
    if inner.startswith("Ptr{"):
        return "ptr:ptr"
    return "ptr:" + {"Float64": "f64", "Int32": "i32", "Int64": "i64", "UInt64": "u64", "UInt8": "any", "Cvoid": "any"}[inner]


def compatible(jc, cc):
    if jc == cc:
        return True
    if jc.startswith("ptr:") and cc.startswith("ptr:"):
        j, c = jc[4:], cc[4:]
        return (j == "any" and c != "ptr") or (c == "any" and j != "ptr")
    return False


def test_every_ccall_of_the_shim_matches_its_prototype():
    calls, protos = julia_ccalls(), header_prototypes()
    assert len(calls) >= 30 and len(protos) >= 60
    problems = []
    for c in calls:
        where = "julia/BEnlsipHIP.jl:%d ccall(:%s)" % (c["line"], c["name"])
        if c["name"] not in protos:
            problems.append("%s: include/benlsip_hip.h declares no such export" % where)
            continue
        p = protos[c["name"]]
        want_ret = "Cstring" if p["ret"] == "cstring" else "Int32"
        if c["ret"] != want_ret:
            problems.append("%s: return type %s, header says %s" % (where, c["ret"], want_ret))
        if len(c["types"]) != len(p["params"]):
            problems.append("%s: %d argument types, the prototype has %d parameters (%s)" % (where, len(c["types"]), len(p["params"]), ", ".join(p["raw"])))
            continue
        if c["nargs"] != len(c["types"]):
            problems.append("%s: %d argument types but %d arguments" % (where, len(c["types"]), c["nargs"]))
        for k, (jt, cc) in enumerate(zip(c["types"], p["params"])):
            if not compatible(julia_class(jt), cc):
                problems.append("%s: argument %d is %s, the prototype has `%s`" % (where, k + 1, jt, p["raw"][k]))
    assert problems == [], "\n".join(problems)
    used = {c["name"] for c in calls}
    # the entry points the shim is built around must all be bound
    for name in ("bh_init", "bh_hess_create", "bh_hess_destroy", "bh_hess_set_mu", "bh_hmul", "bh_vthv", "bh_proj_create", "bh_proj_set_active",
                 "bh_proj_destroy", "bh_project", "bh_pcg", "bh_minor_iterate", "bh_cauchy_step", "bh_grad", "bh_resid_sqnorm",
                 "bh_comm_unique_id", "bh_comm_init", "bh_comm_destroy", "bh_strerror", "bh_last_error_detail"):
        assert name in used, "the shim never calls %s" % name


def test_the_checker_catches_a_transposed_argument():
    """The check must be able to fail: swap an Int64 and a Float64 of bh_pcg's tuple and it has to notice."""
    protos = header_prototypes()
    p = protos["bh_pcg"]
    types = ["Ptr{Cvoid}", "Ptr{Cvoid}", "Ptr{Float64}", "Ptr{Float64}", "Ptr{Float64}", "Float64", "Float64", "Float64",
             "Ptr{Float64}", "Ref{Int32}", "Ref{Int32}", "Ptr{Float64}", "Int64", "Ref{Int32}"]
    assert all(compatible(julia_class(t), c) for t, c in zip(types, p["params"])) and len(types) == len(p["params"])
    bad = list(types)
    bad[7], bad[12] = bad[12], bad[7]
    assert not all(compatible(julia_class(t), c) for t, c in zip(bad, p["params"]))
    assert not compatible(julia_class("Ref{Int32}"), "ptr:f64") and not compatible(julia_class("Ref{Ptr{Cvoid}}"), "ptr:any")
    assert compatible(julia_class("Ptr{Cvoid}"), "ptr:f64") and compatible(julia_class("Ptr{UInt8}"), "ptr:any")


def _strip_julia(src):
    """Comments, string and character literals blanked out (newlines kept), so that keywords and brackets can be counted."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("#=", i):
            j = src.find("=#", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == "#":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith('"""', i):
            j = src.find('"""', i + 3)
            j = n if j < 0 else j + 3
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""' + " " * (j - i - 1)); i = j + 1
        elif c == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src.find("'", i + 2) in (i + 3, i + 4))):
            j = src.find("'", i + 2)
            out.append(" " * (j - i + 1)); i = j + 1            # a character literal ('x', '\n'); a lone ' is the adjoint operator
        else:
            out.append(c); i += 1
    return "".join(out)


def test_julia_shim_blocks_and_brackets_balance():
    """No Julia toolchain exists in this pipeline, so the shim has never been parsed by Julia.  The next best mechanical check: after
    blanking comments and literals, every block opener at bracket depth 0 (function, macro, if, for, while, let, begin, try, struct,
    module, quote, do) has its `end`, `end` inside [...] is an index, and (), [], {} nest and close — a missing `end` or a stray
    bracket anywhere in the file fails here."""
    import re
    src = _strip_julia(open(os.path.join(ROOT, "julia", "BEnlsipHIP.jl")).read())
    openers = {"function", "macro", "if", "for", "while", "let", "begin", "try", "struct", "module", "baremodule", "quote", "do"}
    stack, blocks = [], []
    pairs = {")": "(", "]": "[", "}": "{"}
    line = 1
    for m in re.finditer(r"\n|[A-Za-z_][A-Za-z_0-9!]*|[()\[\]{}]|:[A-Za-z_]\w*", src):
        tok = m.group(0)
        if tok == "\n":
            line += 1
        elif tok in "([{":
            stack.append((tok, line))
        elif tok in ")]}":
            assert stack and stack[-1][0] == pairs[tok], "unbalanced %r in line %d (open: %s)" % (tok, line, stack[-3:])
            stack.pop()
        elif tok.startswith(":"):
            continue                                             # a symbol such as :end or :function
        elif tok in openers and not stack:
            prev = src[max(0, m.start() - 8):m.start()]
            if tok == "struct" and prev.rstrip().endswith("mutable"):
                pass
            blocks.append((tok, line))
        elif tok == "end":
            if any(b == "[" for b, _ in stack):
                continue                                         # a[end]
            assert not stack, "`end` inside brackets in line %d" % line
            assert blocks, "`end` without an opener in line %d" % line
            blocks.pop()
    assert not stack, "unclosed brackets: %s" % stack[-3:]
    assert not blocks, "blocks without `end`: %s" % blocks[-5:]


def test_shim_methods_line_up_with_the_reference_interface():
    """A more specific method only shadows the reference's if Julia's dispatch finds it: same function name, same number of positional
    parameters (a mismatch would not raise — the reference's own CPU method would silently keep running).  Every `function
    BEnlsip.name(...)` / `Base.:*` of the shim against tests/golden/reference_signatures.json (names and arities of the reference's
    functions, written by tests/golden/make_reference_signatures.py): a function of that name with that many positional parameters
    exists, parameter names agree in order, every type annotation is the reference's with T = Float64, and keywords the shim
    accepts are keywords of the reference's method.  Where the
    reference tree is present the fixture is regenerated and must be unchanged."""
    import importlib.util
    import json
    import re
    gold = os.path.join(ROOT, "tests", "golden")
    ref = json.load(open(os.path.join(gold, "reference_signatures.json")))
    spec = importlib.util.spec_from_file_location("mrs", os.path.join(gold, "make_reference_signatures.py"))
    mrs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mrs)
    if os.path.isdir(mrs.REF):
        assert mrs.build() == ref, "fixture out of date: rerun make_reference_signatures.py"
    by_name = {}
    for fname in mrs.FILES:
        for s in ref[fname]:
            by_name.setdefault(s["name"].split(".")[-1].lstrip(":"), []).append(s)
    src = _strip_julia(open(os.path.join(ROOT, "julia", "BEnlsipHIP.jl")).read())
    shim = [s for s in mrs.signatures(src) if s["name"].startswith(("BEnlsip.", "Base."))]
    assert len(shim) >= 13
    for s in shim:
        name = s["name"].split(".")[-1].lstrip(":")
        cands = [r for r in by_name.get(name, []) if len(r["positional"]) == len(s["positional"])]
        assert cands, "shim method %s/%d has no counterpart in the reference (candidates: %s)" % (
            s["name"], len(s["positional"]), [(r["name"], len(r["positional"])) for r in by_name.get(name, [])])
        match = [r for r in cands if r["positional"] == s["positional"]]
        assert match, (s["name"], s["positional"], [r["positional"] for r in cands])
        for kw in s["keywords"]:
            assert any(kw in r["keywords"] for r in cands), (s["name"], kw)
        # each annotation is the reference's with T = Float64 (function-typed parameters F1.. keep their own type variables; an
        # unparametrised `AlHessian` in the reference may be narrowed): more specific, never a different container
        for a, ts, tr in zip(s["positional"], s["types"], match[0]["types"]):
            want = re.sub(r"\bT\b", "Float64", tr)
            got = ts.replace("BEnlsip.", "")
            assert got == want or (want and "{" not in want and got.startswith(want + "{")) or re.fullmatch(r"F\d", got), (s["name"], a, ts, tr)


def test_shim_touches_only_fields_and_enum_members_the_reference_has():
    """Every `H.x` / `lincons.x` the shim reads is a field of the reference's AlHessian / MixedConstraints, and every CG status it
    constructs is a member of the reference's `@enum CG_status` (tests/golden/reference_signatures.json)."""
    import json
    import re
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))
    src = _strip_julia(open(os.path.join(ROOT, "julia", "BEnlsipHIP.jl")).read())
    fields = ref["struct_fields"]
    used_h = set(re.findall(r"\bH\.([A-Za-z_]\w*)", src))
    used_l = set(re.findall(r"\blincons\.([A-Za-z_]\w*)", src))
    assert used_h and used_h <= set(fields["AlHessian"]), (used_h, fields["AlHessian"])
    assert used_l and used_l <= set(fields["MixedConstraints"]), (used_l, fields["MixedConstraints"])
    members = set(ref["CG_status"])
    assert members == {"solved", "bound_hit", "negative_curvature", "max_iter_reached"}
    # the shim converts the library's integer with `BEnlsip.CG_status(code)`: the header's codes must be the enum's positions
    header = open(os.path.join(ROOT, "include", "benlsip_hip.h")).read()
    for pos, member in enumerate(ref["CG_status"]):
        assert re.search(r"#define\s+BH_CG_%s\s+%d\b" % (member.upper(), pos), header), (member, pos)
    assert "BEnlsip.CG_status(status[])" in src.replace(" ", "") or "BEnlsip.CG_status(status[])" in src
    for name in re.findall(r"BEnlsip\.(solved|bound_hit|negative_curvature|max_iter_reached|[a-z_]+_reached)\b", src):
        assert name in members, name


def test_every_reference_function_the_shim_calls_exists_with_that_arity():
    """Calls INTO the reference from the shim (`BEnlsip.cholesky_aug_aat(...)`, `invoke(BEnlsip.cauchy_step, Tuple{...}, ...)`, constructors):
    the name is a function, struct or enum of the reference, and a plain call passes a number of positional arguments that one of its
    methods takes."""
    import importlib.util
    import json
    import re
    gold = os.path.join(ROOT, "tests", "golden")
    ref = json.load(open(os.path.join(gold, "reference_signatures.json")))
    spec = importlib.util.spec_from_file_location("mrs", os.path.join(gold, "make_reference_signatures.py"))
    mrs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mrs)
    arities = {}
    for f in mrs.FILES:
        for s in ref[f]:
            arities.setdefault(s["name"].split(".")[-1], set()).add(len(s["positional"]))
    known = set(arities) | set(ref["struct_fields"]) | {"CG_status"}
    src = _strip_julia(open(os.path.join(ROOT, "julia", "BEnlsipHIP.jl")).read())
    n_calls = 0
    for m in re.finditer(r"(function\s+)?BEnlsip\.([A-Za-z_]\w*!?)(\s*\()?", src):
        is_def, name, call = m.group(1), m.group(2), m.group(3)
        assert name in known, "BEnlsip.%s is not defined by the reference" % name
        if is_def or not call or name not in arities:
            continue                                    # a method definition, a bare reference (invoke, Tuple types) or a constructor
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        pos = src[m.end():i - 1].partition(";")[0]
        n_args = len(mrs.split_args(pos))
        assert n_args in arities[name], "BEnlsip.%s called with %d positional arguments; the reference has %s" % (name, n_args, sorted(arities[name]))
        n_calls += 1
    assert n_calls >= 1
