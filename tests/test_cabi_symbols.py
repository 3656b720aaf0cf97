"""CPU: the C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol declared
in include/*.h with a matching ctypes binding; no compute is called."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(hdr):
    src = open(os.path.join(ROOT, "include", hdr)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(sta_[a-z0-9_]+)\s*\(", src))


def all_exported(path):
    """EVERY defined dynamic symbol of a built library (nm -D; no GPU, nothing is called)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split()}


def exported_symbols(path):
    return {s for s in all_exported(path) if s.startswith("sta_")}


def test_library_builds_and_exports_every_declared_symbol():
    """The product library exports EXACTLY include/sta_mi355.h; the test-hooks build additionally include/sta_mi355_debug.h;
    the ctypes tables mirror the two headers one for one."""
    from vista_slam_amd import build, _lib
    build.build_lib()
    prod, dbg = declared_symbols("sta_mi355.h"), declared_symbols("sta_mi355_debug.h")
    assert prod and dbg and not (prod & dbg), (prod & dbg)
    assert prod == set(_lib.SIGNATURES), (prod ^ set(_lib.SIGNATURES))
    assert dbg == set(_lib.TEST_SIGNATURES), (dbg ^ set(_lib.TEST_SIGNATURES))
    assert exported_symbols(_lib.LIB_PATH) == prod, (exported_symbols(_lib.LIB_PATH) ^ prod)          # no test hooks in the product ABI
    assert exported_symbols(_lib.TEST_LIB_PATH) == prod | dbg, (exported_symbols(_lib.TEST_LIB_PATH) ^ (prod | dbg))
    # -fvisibility=hidden + STA_API: the dynamic symbol table is the header and NOTHING else - no __device_stub__ kernel launch
    # stubs, no helper functions (linker-defined section markers aside)
    linker = {"_init", "_fini", "_edata", "_end", "__bss_start"}
    assert all_exported(_lib.LIB_PATH) - linker == prod, sorted(all_exported(_lib.LIB_PATH) - linker - prod)[:10]
    assert all_exported(_lib.TEST_LIB_PATH) - linker == prod | dbg, sorted(all_exported(_lib.TEST_LIB_PATH) - linker - prod - dbg)[:10]
    lib = _lib.load()
    for name in prod:
        assert hasattr(lib, name), f"{name} declared in include/sta_mi355.h but not exported"
    for name in dbg:
        assert not hasattr(lib, name), f"{name} is a test hook but the product library exports it"
    tlib = _lib.load_test()
    for name in prod | dbg:
        assert hasattr(tlib, name), f"{name} missing from libsta_mi355_test.so"
    assert b"gfx950" in lib.sta_version()


def test_default_config_matches_reference_constructor():
    from vista_slam_amd import _lib
    from vista_slam_amd import weights as W
    lib = _lib.load()
    c = _lib.StaConfig()
    lib.sta_default_config(c)
    f = W.FULL
    assert (c.patch_size, c.enc_embed_dim, c.enc_depth, c.enc_num_heads) == (f.patch_size, f.enc_embed_dim, f.enc_depth, f.enc_num_heads)
    assert (c.dec_embed_dim, c.dec_depth, c.dec_num_heads, c.mlp_ratio) == (f.dec_embed_dim, f.dec_depth, f.dec_num_heads, f.mlp_ratio)
    assert abs(c.rope_base - 100.0) < 1e-6 and abs(c.ln_eps - 1e-6) < 1e-12


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under vista_slam_amd/ may reference it."""
    pkg = os.path.join(ROOT, "vista_slam_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "sta_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vista_slam_amd import _lib
    from vista_slam_amd.sta_frontend import STAFrontend
    with pytest.raises(_lib.StaError):
        STAFrontend()


def test_install_as_reference_registers_the_import_path(tmp_path, monkeypatch):
    """vista_slam_amd.install_as_reference(): `from .sta_model.sta_model import SymmetricTwoViewAssociation as STA` inside a
    package named vista_slam (what vista_slam/slam.py:9 does) resolves to STAFrontend - here with a stand-in package holding a
    stand-in caller (the reference's slam.py needs pypose / cv2 / DBoW3Py, absent everywhere this suite runs).  CPU part: the
    import path; the GPU test test_checkpoint_file_through_the_reference_import_path runs the caller end to end."""
    import importlib
    import sys
    pkg = tmp_path / "vista_slam"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "caller.py").write_text("from .sta_model.sta_model import SymmetricTwoViewAssociation as STA\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == "vista_slam" or k.startswith("vista_slam.")]:
        monkeypatch.delitem(sys.modules, k)
    import vista_slam_amd
    from vista_slam_amd.sta_frontend import STAFrontend
    vista_slam_amd.install_as_reference()
    caller = importlib.import_module("vista_slam.caller")
    assert caller.STA is STAFrontend
    for k in [k for k in sys.modules if k == "vista_slam" or k.startswith("vista_slam.")]:
        monkeypatch.delitem(sys.modules, k)
