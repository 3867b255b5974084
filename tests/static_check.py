"""A small static checker for the GPU-only test modules (no pyflakes in the image): the driver runs `pytest -m gpu -x`, so ONE
NameError / misspelt helper / wrong argument list in code that only ever executes on a GPU box hides every test collected after
it.  Per module, without executing any test body:

* every name a function body loads resolves: a local / enclosing-scope binding, a module-level binding, or a builtin;
* `mod.attr` where `mod` is a module imported at the top of the file: the attribute exists on the imported module;
* calls `f(...)` / `mod.f(...)` / `Class(...)` of plain Python callables with only positional and keyword arguments (no * / **): the
  arguments bind to the callee's signature;
* `x.method(...)` where the function assigned `x = SomeClass(...)` (Detector, StreamedDetector, BoardGather ...) and nothing else: the
  method exists and the arguments bind.
"""
import ast
import builtins
import importlib
import inspect
import os
import symtable
import sys
import types


_MODULE_DUNDERS = {"__file__", "__name__", "__doc__", "__spec__", "__package__", "__builtins__", "__loader__"}


def _module_bindings(tree):
    names = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(node.name)
        elif isinstance(node, ast.Import):
            for a in node.names:
                names.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, ast.ImportFrom):
            for a in node.names:
                names.add(a.asname or a.name)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            names.add(node.id)
        elif isinstance(node, ast.Global):
            names.update(node.names)
    return names


def _undefined_globals(src, path, module_names):
    """Names that function scopes resolve as globals and that nothing at module level binds."""
    bad = []

    def walk(table):
        if table.get_type() == "function":
            for sym in table.get_symbols():
                if sym.is_global() and sym.is_referenced() and sym.get_name() not in module_names \
                        and not hasattr(builtins, sym.get_name()) and sym.get_name() not in _MODULE_DUNDERS:
                    bad.append("%s: function %s() uses undefined name %r" % (os.path.basename(path), table.get_name(), sym.get_name()))
        for ch in table.get_children():
            walk(ch)
    walk(symtable.symtable(src, path, "exec"))
    return bad


def _bind_problem(fn, call):
    if any(isinstance(a, ast.Starred) for a in call.args) or any(k.arg is None for k in call.keywords):
        return None
    try:
        target = fn.__init__ if inspect.isclass(fn) else fn
        if not (inspect.isfunction(target) or inspect.ismethod(target)):
            return None
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return None
    try:
        sig.bind(*[None] * len(call.args), **{k.arg: None for k in call.keywords})
    except TypeError as e:
        return str(e)
    return None


def check_module(path):
    """Returns the list of findings (strings) for one test module; imports it (collecting is what pytest does anyway)."""
    with open(path) as f:
        src = f.read()
    tree = ast.parse(src, path)
    bad = _undefined_globals(src, path, _module_bindings(tree))
    name = os.path.splitext(os.path.basename(path))[0]
    here = os.path.dirname(os.path.abspath(path))
    if os.path.exists(os.path.join(here, "__init__.py")) and os.path.basename(here).isidentifier():
        # a module of a package (img2sgf_amd/pipeline.py): imported under its package name, relative imports work
        parent = os.path.dirname(here)
        if parent not in sys.path:
            sys.path.insert(0, parent)
        mod = importlib.import_module(os.path.basename(here) + "." + name)
    else:
        if here not in sys.path:
            sys.path.insert(0, here)
        mod = importlib.import_module(name)
    base = os.path.basename(path)

    def resolve(node):
        """The object a Name / dotted Attribute chain denotes at module level, or None."""
        if isinstance(node, ast.Name):
            return getattr(mod, node.id, None)
        if isinstance(node, ast.Attribute):
            owner = resolve(node.value)
            if isinstance(owner, types.ModuleType):
                if not hasattr(owner, node.attr):
                    bad.append("%s:%d: module %s has no attribute %r" % (base, node.lineno, owner.__name__, node.attr))
                    return None
                return getattr(owner, node.attr)
            if inspect.isclass(owner):
                return getattr(owner, node.attr, None)
        return None

    def resolve_quiet(node):
        n = len(bad)
        r = resolve(node)
        del bad[n:]
        return r

    for fn_node in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:
        local = {a.arg for a in fn_node.args.args + fn_node.args.kwonlyargs} | {
            n.id for n in ast.walk(fn_node) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)} | {
            (a.asname or a.name).split(".")[0] for n in ast.walk(fn_node) if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names}
        # locals that hold an instance of a known class: `det = Detector(...)`, `a, b = Detector(...), Detector(...)`
        inst = {}
        for node in ast.walk(fn_node):
            if isinstance(node, ast.Assign) and len(node.targets) == 1:
                pairs = []
                if isinstance(node.targets[0], ast.Name):
                    pairs = [(node.targets[0], node.value)]
                elif isinstance(node.targets[0], ast.Tuple) and isinstance(node.value, ast.Tuple) and len(node.targets[0].elts) == len(node.value.elts):
                    pairs = [(t, v) for t, v in zip(node.targets[0].elts, node.value.elts) if isinstance(t, ast.Name)]
                for t, v in pairs:
                    cls = resolve_quiet(v.func) if isinstance(v, ast.Call) else None
                    if inspect.isclass(cls) and cls.__module__.startswith(("img2sgf_amd", "test", "parity")):
                        inst.setdefault(t.id, set()).add(cls)
                    elif t.id in inst:
                        inst[t.id].add(None)                       # rebound to something else: leave it alone
        for node in ast.walk(fn_node):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id not in local:
                resolve(node)
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) \
                    and len(inst.get(node.func.value.id, ())) == 1 and None not in inst[node.func.value.id]:
                cls = next(iter(inst[node.func.value.id]))
                if not hasattr(cls, node.func.attr):
                    bad.append("%s:%d: %s (a %s) has no method %r" % (base, node.lineno, node.func.value.id, cls.__name__, node.func.attr))
                else:
                    meth = getattr(cls, node.func.attr)
                    if inspect.isfunction(meth) and not any(isinstance(a, ast.Starred) for a in node.args) \
                            and all(k.arg is not None for k in node.keywords):
                        try:
                            inspect.signature(meth).bind(None, *[None] * len(node.args), **{k.arg: None for k in node.keywords})
                        except TypeError as e:
                            bad.append("%s:%d: call of %s.%s: %s" % (base, node.lineno, cls.__name__, node.func.attr, e))
            if isinstance(node, ast.Call):
                root = node.func
                while isinstance(root, ast.Attribute):
                    root = root.value
                if isinstance(root, ast.Name) and root.id not in local:
                    callee = resolve(node.func)
                    if callee is not None and not isinstance(callee, types.ModuleType):
                        msg = _bind_problem(callee, node)
                        if msg:
                            bad.append("%s:%d: call of %s: %s" % (base, node.lineno, ast.unparse(node.func), msg))
    return sorted(set(bad))
