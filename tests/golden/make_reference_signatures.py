"""Writes tests/golden/reference_signatures.json: for every `function name(...)` in the reference's live source files, its name and
its first line, the number, names and type annotations of its POSITIONAL parameters (keywords after `;` listed separately); the field names of its structs, the members of `CG_status` and the length of each file.  Data about the reference's interface — what
the Julia shim's more specific methods must line up with — not its source.  Run here (needs /root/reference); the fixture is committed.
    python tests/golden/make_reference_signatures.py"""
import json
import os
import re

REF = "/root/reference/src"
FILES = ["basic_tralcnlss.jl", "polyhedral_constraints.jl"]


def split_args(s):
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


def signatures(text):
    sigs = []
    for m in re.finditer(r"^function\s+([A-Za-z_][\w!.:*]*)\s*\(", text, re.M):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        inner = text[m.end():i - 1]
        pos, _, kw = inner.partition(";")
        names = lambda lst: [re.split(r"::|=", a)[0].strip() for a in lst]
        types = lambda lst: [re.sub(r"\s+", "", re.split(r"=", a.split("::", 1)[1])[0]) if "::" in a else "" for a in lst]
        sigs.append({"name": m.group(1), "positional": names(split_args(pos)), "types": types(split_args(pos)), "keywords": names(split_args(kw)),
                     "line": text.count("\n", 0, m.start()) + 1})
    return sigs


def struct_fields(text):
    """{struct name: [field names]} of the `struct ... end` blocks (plain `name::Type` lines)."""
    out = {}
    for m in re.finditer(r"^(?:mutable\s+)?struct\s+([A-Za-z_]\w*)[^\n]*\n(.*?)^end", text, re.M | re.S):
        out[m.group(1)] = [ln.split("::")[0].strip() for ln in m.group(2).splitlines() if "::" in ln and not ln.strip().startswith("#")]
    return out


def build():
    out = {}
    for f in FILES:
        out[f] = signatures(open(os.path.join(REF, f)).read())
    out["struct_fields"] = {}
    for f in FILES:
        out["struct_fields"].update(struct_fields(open(os.path.join(REF, f)).read()))
    enum = re.search(r"@enum\s+CG_status\s+([^\n]+)", open(os.path.join(REF, FILES[0])).read())
    out["CG_status"] = enum.group(1).split()
    out["n_lines"] = {f: open(os.path.join(REF, f)).read().count("\n") for f in FILES}
    return out


def main():
    out = build()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_signatures.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print({f: (len(v) if hasattr(v, "__len__") else v) for f, v in out.items()})
    print(out["struct_fields"], out["CG_status"])


if __name__ == "__main__":
    main()
