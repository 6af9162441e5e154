"""bench.py's roofline.traffic comes from a committed PMC summary (PMC passes cannot run inside the bench process); it is
quoted only when that summary was measured on the same build of the network kernels as the loaded library (CPU test)."""
import json
import os
import sys

from conftest import REPO

sys.path.insert(0, REPO)


def _summary(tmp_path, nn_hash):
    p = os.path.join(tmp_path, "pmc.json")
    json.dump({"build": {"src": "aaaa", "nn": nn_hash}, "traffic_bytes_per_launch": 200e6, "evals_per_launch": 5000.0,
               "mfma_busy_frac": 0.7, "steps": 20, "warmup": 5}, open(p, "w"))
    return p


def test_traffic_is_scaled_when_the_build_matches(tmp_path):
    import bench
    tr, note, busy = bench.tower_traffic(_summary(str(tmp_path), "1234abcd"), "1234abcd", 2500.0)
    assert abs(tr - 100e6) < 1 and busy == 0.7 and "same build nn=1234abcd" in note


def test_traffic_is_withheld_when_the_kernels_changed(tmp_path):
    import bench
    tr, note, busy = bench.tower_traffic(_summary(str(tmp_path), "1234abcd"), "ffff0000", 2500.0)
    assert tr is None and busy is None and "withheld" in note and "ffff0000" in note


def test_traffic_is_null_without_a_summary(tmp_path):
    import bench
    tr, note, _ = bench.tower_traffic(os.path.join(str(tmp_path), "missing.json"), "x", 1.0)
    assert tr is None and "no PMC summary" in note


def test_loaded_library_reports_the_hash_of_the_sources_in_the_tree():
    """dbaz_build_info() of the in-tree library = the hash build.py computes over csrc/ now (the library is current)."""
    from dotsboxesaz_amd import _lib, build
    build.build()
    info = _lib.build_info()
    assert info["src"] == build.source_hash() and info["nn"] == build.source_hash(build.NN_SOURCES)
