#!/usr/bin/env python
"""Rough undefined-name check (no pyflakes in this image): names loaded anywhere in a file that are never bound in it
(assignment, argument, import, def/class, comprehension / with / except / for target) and are not builtins."""
import ast
import builtins
import sys


def check(path):
    tree = ast.parse(open(path).read(), path)
    bound, loads = set(dir(builtins)) | {"__file__", "__name__"}, []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name):
            (loads if isinstance(n.ctx, ast.Load) else bound).append(n) if isinstance(n.ctx, ast.Load) else bound.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
        elif isinstance(n, ast.arg):
            bound.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
    return sorted({(n.id, n.lineno) for n in loads if n.id not in bound})


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for name, line in check(p):
            print("%s:%d: undefined name %s" % (p, line, name))
            bad += 1
    sys.exit(1 if bad else 0)
