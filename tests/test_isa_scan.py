"""CPU: static check of the built code object for loads queued behind stores (tools/isa_serial_scan.py).

On gfx9 global loads and stores retire through one in-order counter (vmcnt), so the data of a load issued behind a store cannot be
used before that store has been acknowledged.  Round 3 found the in-place residual epilogue compiled to `store, load,
s_waitcnt vmcnt(0)` sixteen times per 32x32 tile (82-92 such waits in the kernel; +4.1 % on the whole step once the loads were
issued first).  The scan counts, in program order, the waits on a load that was issued while an earlier store was still pending;
this test keeps the hot kernels from regressing to the per-element form (what is left are the per-tile bias / table loads)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hot_epilogues_do_not_queue_loads_behind_stores():
    import pytest
    import kernel_resources as kr
    if not os.path.exists(kr.LIB):
        pytest.skip("libsta_mi355.so not built here (python -m vista_slam_amd.build)")
    if not os.path.exists(os.path.join(kr.LLVM, "llvm-objdump")):
        pytest.skip("ROCm LLVM tools (llvm-objdump) not installed on this box")
    import isa_serial_scan
    stats = isa_serial_scan.scan()
    assert len(stats) > 100, "code object not parsed"
    resid = {k: v for k, v in stats.items() if k.startswith("_Z12gemm2_kernel") and "ELi0ELi5E" in k}      # dense, EPI_F32R
    assert len(resid) >= 6, sorted(resid)
    for k, v in resid.items():
        assert v["serial"] <= 16, (k, v)             # 10 today; 82-92 in the per-element form
    conv = {k: v for k, v in stats.items() if k.startswith("_Z13conv3h_kernel")}
    assert conv and all(v["serial"] <= 8 for v in conv.values()), conv     # residual planes of the DPT convolutions: 5 today
    ln = {k: v for k, v in stats.items() if "ln_kernel" in k}
    assert ln and all(v["serial"] <= 2 for v in ln.values()), ln           # affine parameters loaded with the row
