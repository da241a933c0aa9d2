"""The oracle is test infrastructure: nothing the product ships may import, link or execute it (or any CPU fallback).
Allowed users: tests/, __graft_entry__.smoke() (the checker of the smoke run) and bench.py's cpu_baseline() leg."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT_DIRS = ("ship_sim_gym_amd", "ship_gym", "train")


def _py_files(d):
    for base, dirs, files in os.walk(os.path.join(ROOT, d), followlinks=False):
        dirs[:] = [x for x in dirs if x != "__pycache__"]
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def _imports(path):
    tree = ast.parse(open(path).read(), path)
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom):
            yield node.module or ""


def test_product_python_never_touches_the_oracle():
    for d in PRODUCT_DIRS:
        for path in _py_files(d):
            for mod in _imports(path):
                assert mod.split(".")[0] != "oracle", "%s imports %s" % (path, mod)
            src = open(path).read()
            assert "libssg_oracle" not in src and "oracle/_build" not in src and "oracle/_ref" not in src, path


def test_product_native_sources_do_not_reference_the_oracle():
    csrc = os.path.join(ROOT, "ship_sim_gym_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".cpp", ".h")) or f == "Makefile":
            src = open(os.path.join(csrc, f)).read()
            # (comments may NAME oracle files as the checker; no include, no symbol, no path into oracle/ may be used)
            assert not re.search(r'#\s*include\s*[<"][^>"]*(oracle|ssg_oracle|ssg_dynamics\.c)', src), f
            assert "ora_" not in re.sub(r"//.*|/\*.*?\*/", "", src, flags=re.S), f
            if f == "Makefile":
                assert "oracle" not in src, f


def test_bench_and_entry_use_the_oracle_only_as_checker():
    bench = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(bench)
    users = [fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef)
             and any(m.split(".")[0] == "oracle" for m in _imports_of(fn))]
    assert users == ["cpu_baseline"], users
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    tree = ast.parse(entry)
    users = [fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef)
             and any(m.split(".")[0] == "oracle" for m in _imports_of(fn))]
    assert set(users) <= {"smoke", "build", "_build_oracle"}, users
    # module level: no oracle import outside functions in either file
    for src in (bench, entry):
        for node in ast.parse(src).body:
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""]
                assert all(m.split(".")[0] != "oracle" for m in mods)


def _imports_of(fn):
    for node in ast.walk(fn):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom):
            yield node.module or ""


def test_product_fails_loudly_without_the_hip_library_or_a_device(tmp_path):
    """No CPU fallback: a missing libshipsim.so is a ShipSimError naming the build command, and ShipVecEnv on a box without
    a HIP device refuses (checked in a child process so that this process's loaded library is left alone)."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ["SSG_LIB_PATH"] = %r
from ship_sim_gym_amd import _native as N
try:
    N.lib()
except N.ShipSimError as e:
    assert "no CPU fallback" in str(e) and "build" in str(e), str(e)
    print("MISSING_LIB_OK")
import torch
if not torch.cuda.is_available():
    os.environ.pop("SSG_LIB_PATH")
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    try:
        ShipVecEnv(4)
    except N.ShipSimError as e:
        assert "no CPU fallback" in str(e), str(e)
        print("NO_DEVICE_OK")
else:
    print("NO_DEVICE_OK")  # (a GPU box: nothing to refuse)
''' % (ROOT, str(tmp_path / "nowhere" / "libshipsim.so"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "MISSING_LIB_OK" in p.stdout and "NO_DEVICE_OK" in p.stdout, p.stdout
