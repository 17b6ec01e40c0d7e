#!/usr/bin/env python3
"""Golden vectors for the one piece of the drift schedule the reference snapshot DOES hold: `get_drift_deferential_cosine(t, T)`
(models/drift_noise_model.py:10-16), the per-step increment of the half-cosine drift level.  The module itself cannot be imported
(clip, ema_pytorch, .modules.* are absent), so the function's own definition is taken out of the source file's syntax tree and
evaluated with numpy -- dev container only; the output is data: tests/golden/drift_cosine_golden.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_drift_cosine.py
"""
import ast
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

REF = os.environ.get("IDIFF_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "drift_cosine_golden.npz")


def load_function():
    path = os.path.join(REF, "models", "drift_noise_model.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_drift_deferential_cosine"]
    assert len(fn) == 1
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns["get_drift_deferential_cosine"]


def main():
    f = load_function()
    out = {}
    for T in (50, 100, 1000):
        t = np.arange(T, dtype=np.float64)
        out[f"T{T}/increment"] = np.asarray(f(t, T), dtype=np.float64)            # level(t+1) - level(t), t = 0..T-1
        out[f"T{T}/increment_int_t"] = np.asarray([f(int(k), T) for k in range(T)], dtype=np.float64)
    np.savez(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
