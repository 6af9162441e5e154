"""The drop-in boundary is a C ABI: examples/selfplay_demo.c uses include/dbaz.h and libdbaz_hip.so only
(no Python, no torch).  It must compile and link as plain C11 everywhere, and play games on a GPU box."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(REPO, "dotsboxesaz_amd")


def _build(tmp_path):
    exe = str(tmp_path / "selfplay_demo")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(REPO, "include"),
           os.path.join(REPO, "examples", "selfplay_demo.c"), "-o", exe, "-L" + LIBDIR, "-ldbaz_hip",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_example_compiles_and_links(tmp_path):
    from dotsboxesaz_amd import build
    build.build()
    _build(tmp_path)


@pytest.mark.gpu
def test_c_example_plays_games(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "48"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("games 48 rows ")
    rows = int(r.stdout.split()[3])
    assert rows >= 48 * 8
    assert "mean(sum pi) 1.000000" in r.stdout
