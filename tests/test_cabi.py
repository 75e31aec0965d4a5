"""The C-ABI library loads (no GPU needed) and exports every symbol include/pointseg.h declares; the ctypes
prototype table covers exactly that set."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pointseg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", src))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), "libpointseg_hip.so does not export %s" % n


def test_prototype_table_matches_header():
    from point_unet_amd import _lib
    assert set(_lib.PROTOTYPES) == _declared()


def test_version_and_error_strings(lib):
    assert lib.ps_version().decode().startswith("pointseg-hip")
    assert lib.ps_knn_batch(None, None, None, 1, 1, 1, 3, 16, None, 0) != 0   # argument check fails before any GPU call
    assert b"NULL" in lib.ps_last_error()


def test_debug_doors_live_in_their_own_library(lib, dbg):
    """The ps_debug_* test doors are not in the product library (and not in its header): csrc/debug_hooks.h / libpointseg_debug.so."""
    for name in ("ps_debug_knn_host", "ps_debug_kdtree_host", "ps_debug_kdtree_device", "ps_debug_pack_weights", "ps_debug_pack_b3"):
        assert not hasattr(lib, name) and hasattr(dbg, name)
    assert "ps_debug" not in open(os.path.join(ROOT, "include", "pointseg.h")).read()
    assert dbg.ps_debug_knn_host(None, None, 1, 1, 1, 16, None) != 0 and b"NULL" in lib.ps_last_error()


def test_product_never_imports_the_oracle():
    """The product package and bench's timed path must not route through oracle/ (parity rule)."""
    pkg = os.path.join(ROOT, "point-unet_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle's", "").replace("oracle/", "ORACLE_DIR_MENTION") or \
                    "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)
                assert "from oracle" not in txt and "import oracle" not in txt and "liboracle" not in txt, os.path.join(dp, f)
