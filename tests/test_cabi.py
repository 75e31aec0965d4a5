"""The C-ABI library loads (no GPU needed) and exports every symbol include/pointseg.h declares; the ctypes
prototype table covers exactly that set."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("pointseg.h", "pointseg_train_ops.h")  # the stable surface of the path / the op-level kernels of the training step


def _declared(headers=HEADERS):
    names = set()
    for hname in headers:
        src = open(os.path.join(ROOT, "include", hname)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", src))
    return names


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), "libpointseg_hip.so does not export %s" % n


def test_prototype_table_matches_header():
    from point_unet_amd import _lib
    assert set(_lib.PROTOTYPES) == _declared()


def test_the_stable_header_stays_small():
    """include/pointseg.h is the surface a reference maintainer binds (context, knn, pyramid, grid, volume, the forward, the
    six Network.* ops, the training step behind one call); the ~60 op-level kernels of the training step live in pointseg_train_ops.h
    and none of them leaks back."""
    stable, ops = _declared(("pointseg.h",)), _declared(("pointseg_train_ops.h",))
    assert not (stable & ops)
    assert len(stable) <= 60, sorted(stable)
    assert {"ps_knn_batch", "ps_pyramid_build", "ps_grid_subsample", "ps_volume_to_cloud", "ps_randla_forward", "ps_randla_train_step",
            "ps_op_gather_neighbour", "ps_op_relative_pos_encoding", "ps_op_att_pool", "ps_op_random_sample", "ps_op_nearest_interpolation"} <= stable
    assert all(n.startswith("ps_op_") or n == "ps_set_train_act_bf16" for n in ops)  # (the one setter that only concerns those kernels)


def test_version_and_error_strings(lib):
    assert lib.ps_version().decode().startswith("pointseg-hip")
    hdr = open(os.path.join(ROOT, "include", "pointseg.h")).read()
    assert lib.ps_abi_version() == int(re.search(r"#define PS_ABI_VERSION (\d+)", hdr).group(1))  # header, library and ctypes table agree
    assert lib.ps_knn_batch(None, None, None, 1, 1, 1, 3, 16, None, 0) != 0   # argument check fails before any GPU call
    assert b"NULL" in lib.ps_last_error()


def test_debug_doors_live_in_their_own_library(lib, dbg):
    """The ps_debug_* test doors are not in the product library (and not in its header): csrc/debug_hooks.h / libpointseg_debug.so."""
    for name in ("ps_debug_knn_host", "ps_debug_kdtree_host", "ps_debug_kdtree_device", "ps_debug_pack_weights", "ps_debug_pack_b3"):
        assert not hasattr(lib, name) and hasattr(dbg, name)
    for hname in HEADERS:
        assert "ps_debug" not in open(os.path.join(ROOT, "include", hname)).read()
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
