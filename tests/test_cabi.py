"""The C-ABI shared library: loads on a CPU-only host and exports exactly what
include/hipeig.h declares and what the ctypes binding expects.  No compute calls."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from eigensolvers_amd import _lib


def _header_symbols():
    text = open(os.path.join(REPO, "include", "hipeig.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hipeig_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.hipeig_last_error() is not None


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"{name} declared in hipeig.h but not exported"


def test_binding_covers_the_header():
    declared = set(_header_symbols()) - {"hipeig_last_error"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_opaque_types_are_not_exposed():
    text = open(os.path.join(REPO, "include", "hipeig.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "extern \"C\"" in text


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipEigError):
        _lib.load()


def test_header_is_plain_c_and_links(tmp_path):
    """include/hipeig.h compiles as C (not only C++) and a C program that references every declared
    entry point links against libhipeig.so - the boundary carries no C++ or torch types."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    names = _header_symbols()
    src = tmp_path / "use_all.c"
    body = "\n".join(f"  table[{i}] = (void*)&{n};" for i, n in enumerate(names))
    src.write_text('#include "hipeig.h"\n#include <stdio.h>\nint main(void) {\n'
                   f"  void* table[{len(names)}];\n{body}\n"
                   f'  printf("%d %p\\n", {len(names)}, table[0]);\n  return 0;\n}}\n')
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "use_all"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe),
           "-L", lib_dir, "-l:libhipeig.so", f"-Wl,-rpath,{lib_dir}", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
