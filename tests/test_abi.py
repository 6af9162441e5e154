"""The C-ABI library loads and exports every symbol include/dbaz.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from dotsboxesaz_amd import _lib


def header_symbols():
    src = open(os.path.join(REPO, "include", "dbaz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dbaz_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from dotsboxesaz_amd import build
        build.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(L, name), name
    assert _lib.load().dbaz_version() == _lib.ABI_VERSION == 3


def test_struct_sizes_match_header():
    # dbaz_config / dbaz_counters layouts as the C compiler sees them
    assert ctypes.sizeof(_lib.Config) == 232
    assert ctypes.sizeof(_lib.Counters) == 144
    assert ctypes.sizeof(_lib.NetTensors) == 24 * 8 and ctypes.sizeof(_lib.NetRunning) == 10 * 8


def test_struct_sizes_as_gcc_sees_the_header(tmp_path):
    """include/dbaz.h compiled as plain C: sizeof of every struct that crosses the boundary equals the ctypes mirror's."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "dbaz.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(dbaz_config), '
                   'sizeof(dbaz_counters), sizeof(dbaz_net_tensors), sizeof(dbaz_net_running)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout.split()
    assert [int(v) for v in out] == [ctypes.sizeof(_lib.Config), ctypes.sizeof(_lib.Counters), ctypes.sizeof(_lib.NetTensors),
                                     ctypes.sizeof(_lib.NetRunning)]


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dotsboxesaz_amd.engine import Engine
    with pytest.raises(_lib.DbazError) as ei:
        Engine(3, 3, 4)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_bad_config_rejected_before_touching_the_device():
    from dotsboxesaz_amd.engine import Engine
    with pytest.raises(_lib.DbazError):
        Engine(12, 12, 4)  # A = 338 > 256
    with pytest.raises(_lib.DbazError):
        Engine(3, 3, 0)


def _header_struct_fields(name):
    src = open(os.path.join(REPO, "include", "dbaz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        for part in decl.split(","):
            m = re.search(r"([a-z0-9_]+)\s*$", part.strip())
            if m and part.strip():
                fields.append(m.group(1))
    return fields


def test_net_structs_field_order_matches_header():
    """dbaz_net_tensors / dbaz_net_running are structs of pointers: the ctypes mirrors must list them in the header's order."""
    assert _header_struct_fields("dbaz_net_tensors") == [f for f, _ in _lib.NetTensors._fields_]
    assert _header_struct_fields("dbaz_net_running") == [f for f, _ in _lib.NetRunning._fields_]
    from dotsboxesaz_amd import train_tower
    assert list(train_tower._NET_SINGLE) + list(train_tower._NET_BLOCK) == sorted(
        [f for f, _ in _lib.NetTensors._fields_], key=lambda f: (f.startswith("blk_"), [g for g, _ in _lib.NetTensors._fields_].index(f)))
