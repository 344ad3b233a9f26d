"""Writes tests/golden/reference_signatures.json: the parameter lists of the reference's module-level functions.

Test infrastructure (never imported by the product).  The reference's modules are read as TEXT and parsed with ``ast``
(nothing is imported or executed): for every module-level ``def`` of the four modules the drop-in mirrors, the
positional parameter names in order, how many of the trailing ones have defaults, and the keyword-only names.  The
fixture is data about the boundary (names), not source text.  ``tests/test_reference_signatures.py`` holds every
same-named function of ``aind_smartspim_destripe_amd`` to these lists.  Also recorded: the keyword sets of the two
call sites the chunk-map entry points must accept (``run_capsule.py:394-403`` -> ``destripe_channel``,
``zarr_destriper.py:1252-1267`` -> ``destripe_zarr``).

Run:  python oracle/make_reference_signatures.py   (needs /root/reference; not available on the GPU box)
"""

import ast
import json
import os

REF = "/root/reference/code"
MODULES = ["zarr_destriper", "filtering", "destriper", "readers"]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_signatures.json")


def signatures(path):
    tree = ast.parse(open(path).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            a = node.args
            out[node.name] = {
                "positional": [x.arg for x in a.posonlyargs + a.args],
                "n_defaults": len(a.defaults),
                "kwonly": [x.arg for x in a.kwonlyargs],
                "vararg": a.vararg.arg if a.vararg else None,
                "kwarg": a.kwarg.arg if a.kwarg else None,
                "line": node.lineno,
            }
    return out


def call_keywords(path, func_attr):
    """Keyword names of every call of ``<anything>.func_attr(...)`` / ``func_attr(...)`` in ``path``."""
    tree = ast.parse(open(path).read())
    found = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call):
            f = node.func
            name = f.attr if isinstance(f, ast.Attribute) else getattr(f, "id", None)
            if name == func_attr:
                found.append({"line": node.lineno, "n_positional": len(node.args),
                              "keywords": [k.arg for k in node.keywords]})  # fmt: skip
    return found


def main():
    data = {"source": "aind-smartspim-destripe @ /root/reference (2025-05-23), parsed with ast", "modules": {}, "call_sites": {}}
    for m in MODULES:
        data["modules"][m] = signatures(os.path.join(REF, "aind_smartspim_destripe", m + ".py"))
    data["call_sites"]["destripe_channel"] = call_keywords(os.path.join(REF, "run_capsule.py"), "destripe_channel")
    data["call_sites"]["destripe_zarr"] = call_keywords(
        os.path.join(REF, "aind_smartspim_destripe", "zarr_destriper.py"), "destripe_zarr")
    data["call_sites"]["compute_multiscale"] = call_keywords(
        os.path.join(REF, "aind_smartspim_destripe", "zarr_destriper.py"), "compute_multiscale")
    data["call_sites"]["execute_worker"] = call_keywords(
        os.path.join(REF, "aind_smartspim_destripe", "zarr_destriper.py"), "execute_worker")
    data["call_sites"]["filter_stripes"] = call_keywords(
        os.path.join(REF, "aind_smartspim_destripe", "zarr_destriper.py"), "filter_stripes") + call_keywords(
        os.path.join(REF, "aind_smartspim_destripe", "destriper.py"), "filter_stripes")
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
