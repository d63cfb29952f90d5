"""Poor man's pyflakes (no network, no pyflakes here): report names that are loaded but never bound in a
module -- catches typos before a GPU-minute is spent on them.  usage: python tools/lint_names.py file.py ..."""
import ast
import builtins
import sys


def check(path):
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__path__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(node.name)
            if not isinstance(node, ast.ClassDef):
                a = node.args
                for arg in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                    bound.add(arg.arg)
        elif isinstance(node, ast.Lambda):
            a = node.args
            for arg in a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                bound.add(arg.arg)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            bound.add(node.id)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            for al in node.names:
                bound.add((al.asname or al.name).split(".")[0])
        elif isinstance(node, ast.ExceptHandler) and node.name:
            bound.add(node.name)
    bad = sorted({(n.lineno, n.id) for n in ast.walk(tree)
                  if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound})
    for line, name in bad:
        print(f"{path}:{line}: undefined name {name!r}")
    return len(bad)


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
