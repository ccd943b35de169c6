"""Undefined-name check for a Python file without running it (no linters in this image): every name a function reads that is
neither local to it (or to an enclosing function), nor defined at module level, nor a builtin.   python tools/check_names.py FILE..."""
import ast, builtins, symtable, sys


def check(path):
    src = open(path).read()
    tree = ast.parse(src, path)
    top = symtable.symtable(src, path, "exec")
    module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    bad = []

    def walk(tab):
        for child in tab.get_children():
            for s in child.get_symbols():
                if s.is_global() and s.is_referenced() and not s.is_assigned():
                    n = s.get_name()
                    if n not in module_names and not hasattr(builtins, n) and n not in ('__file__', '__name__'):
                        bad.append((child.get_name(), child.get_lineno(), n))
            walk(child)
    walk(top)
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for fn, line, name in check(p):
            print("%s: function %s (line %d) reads undefined name %r" % (p, fn, line, name))
            rc = 1
    sys.exit(rc)
