"""CPU: the C-ABI library loads without a GPU and exports exactly what include/blgnn.h
declares; the ctypes table in blgnn/_lib.py agrees with the header."""

import ctypes
import os
import re

import conftest  # noqa: F401  (sys.path)
from blgnn import _lib

HEADER = os.path.join(conftest.ROOT, "include", "blgnn.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int64_t|int|const\s+char\s*\*)\s*(blgnn_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args == "void" else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_loads_and_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 19
    for name in decl:
        assert hasattr(lib, name), "libblgnn.so does not export %s" % name


def test_ctypes_table_matches_header():
    decl = _declared()
    table = dict(_lib.SIGNATURES)
    for name in _lib.LIBRARY_SYMBOLS:
        assert name in decl
    compute = {k: v for k, v in decl.items() if k not in _lib.LIBRARY_SYMBOLS}
    assert set(compute) == set(table)
    for name, nargs in compute.items():
        assert len(table[name]) == nargs, name


def test_version_and_device_probe_without_gpu():
    lib = _lib.load()
    assert lib.blgnn_version() >= 100
    assert _lib.device_count() >= 0          # 0 in the CPU container, never an error


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from blgnn import ops
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.spmm_sum(None, torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="no CPU"):
        blgnn_layer = __import__("blgnn").NodeEmbedding("gin", 4, 4, "relu", False, False)
        blgnn_layer(torch.zeros(2, 4), None, None)


def test_product_never_imports_oracle():
    pkg = os.path.join(conftest.ROOT, "bilevel-graph-neural-network_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".sh")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_every_ctypes_call_site_passes_the_declared_number_of_arguments():
    """a call site that lags behind a signature change only fails on a GPU box (ctypes raises at call time); count the
    arguments of every ``call("blgnn_*", ...)`` in the package statically (calls that splat a tuple are skipped)"""
    import ast
    import glob
    import os
    from blgnn import _lib
    pkg = os.path.dirname(os.path.abspath(_lib.__file__))
    checked = 0
    for path in glob.glob(os.path.join(pkg, "*.py")):
        for node in ast.walk(ast.parse(open(path).read())):
            if not (isinstance(node, ast.Call) and getattr(node.func, "id", getattr(node.func, "attr", None)) == "call"):
                continue
            if not node.args or not isinstance(node.args[0], ast.Constant) or not str(node.args[0].value).startswith("blgnn_"):
                continue
            if any(isinstance(a, ast.Starred) for a in node.args):
                continue
            name = node.args[0].value
            assert len(node.args) - 1 == len(_lib.SIGNATURES[name]), (os.path.basename(path), node.lineno, name)
            checked += 1
    assert checked > 40
