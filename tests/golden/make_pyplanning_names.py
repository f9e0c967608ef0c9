#!/usr/bin/env python3
"""Extracts the NAMES the reference's Python module binds (interfaces/python/src/pyplanning.cpp): for every class_ / enum_ the
Python-side class name and the names given to .def / .def_static / .def_readonly / .def_readwrite / .def_property* / .value, plus
the module-level m.def names.  Output: tests/golden/pyplanning_bound_names.json -- names only (data), no source text.
Run in the dev container (needs /root/reference); tests/test_pyplanning_surface.py diffs the product's module against the list."""
import json
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/interfaces/python/src/pyplanning.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyplanning_bound_names.json")


def main():
    text = open(SRC).read()
    text = re.sub(r"//[^\n]*", "", text)  # line comments (the banner art contains quotes)
    # statements: from a class_< / enum_< opener to the terminating ';'
    out = {"module": sorted(set(re.findall(r"\bm\.def\(\s*\"([^\"]+)\"", text))), "classes": {}}
    for m in re.finditer(r"\b(class_|enum_)<[^;]*?>\s*\(\s*m\s*,\s*\"([^\"]+)\"[^;]*;", text, flags=re.S):
        kind, name, body = m.group(1), m.group(2), m.group(0)
        # template arguments of .def<...>( contain parentheses and angle brackets but never a quote or a line break
        members = re.findall(r"\.(def_static|def_readonly|def_readwrite|def_property_readonly|def_property|def|value)\b[^\"\n]*?\(\s*\"([^\"]+)\"", body)
        out["classes"][name] = {"kind": "enum" if kind == "enum_" else "class", "members": sorted(set(n for _, n in members))}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print(OUT, len(out["classes"]), "classes,", sum(len(c["members"]) for c in out["classes"].values()), "member names")


if __name__ == "__main__":
    main()
