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
