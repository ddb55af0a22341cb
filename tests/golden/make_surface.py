#!/usr/bin/env python3
"""Record the module surface a driver written against the reference relies on -> tests/golden/surface.json.

Run in the build container only (the reference tree does not exist on the GPU box):
    python /root/repo/tests/golden/make_surface.py

The reference's files are PARSED (ast), never imported or executed.  What is stored is names and numbers only:
  defs   per module of the reference's `src` package: every module-level function with its parameter names, the
         number of parameters that have defaults and the defaults themselves when they are plain literals;
  uses   every `module.attribute` expression found in custom_slam.py and in src/*.py for the six modules the driver
         imports (display, epipolar, logger, plot, triangulation, util), with the shape of each call: number of
         positional arguments and keyword names.
No source text of the reference is stored.
"""
import ast
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
MODULES = ("display", "epipolar", "logger", "plot", "triangulation", "util")


def literal(node):
    try:
        v = ast.literal_eval(node)
    except Exception:
        return {"literal": False}
    if isinstance(v, tuple):
        v = list(v)
    return {"literal": True, "value": v}


def definitions(path):
    tree = ast.parse(open(path).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            a = node.args
            out[node.name] = {
                "params": [p.arg for p in a.posonlyargs + a.args],
                "n_defaults": len(a.defaults),
                "defaults": [literal(d) for d in a.defaults],
                "varargs": a.vararg is not None,
                "kwargs": a.kwarg is not None,
                "line": node.lineno,
            }
    return out


def uses(path):
    tree = ast.parse(open(path).read())
    calls = {}
    # call sites first, so that the attribute walk below can tell plain reads from calls
    called = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) \
                and node.func.value.id in MODULES:
            key = f"{node.func.value.id}.{node.func.attr}"
            called.add(id(node.func))
            shape = {"n_positional": len(node.args), "keywords": sorted(k.arg for k in node.keywords if k.arg),
                     "star": any(isinstance(x, ast.Starred) for x in node.args), "line": node.lineno}
            calls.setdefault(key, {"calls": [], "reads": 0})["calls"].append(shape)
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in MODULES \
                and id(node) not in called:
            calls.setdefault(f"{node.value.id}.{node.attr}", {"calls": [], "reads": 0})["reads"] += 1
    return calls


def main():
    out = {"defs": {}, "uses": {}}
    for m in MODULES:
        out["defs"][m] = definitions(os.path.join(REF, "src", m + ".py"))
    files = ["custom_slam.py"] + [os.path.join("src", m + ".py") for m in MODULES]
    for f in files:
        u = uses(os.path.join(REF, f))
        if u:
            out["uses"][f] = u
    with open(os.path.join(HERE, "surface.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    n_defs = sum(len(v) for v in out["defs"].values())
    n_uses = sum(len(v) for v in out["uses"].values())
    print(f"surface.json: {n_defs} function definitions, {n_uses} distinct module.attribute uses")


if __name__ == "__main__":
    main()
