"""The C-ABI library loads here (no GPU) and exports every symbol include/gstfwd.h declares; without a
device every fill fails loudly; the product package never touches oracle/ or /root/reference."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_fixture, plan_from_fixture
from pygsti_amd import _lib


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gstfwd.h")).read()
    declared = set(re.findall(r"\b(gst_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.gst_version()


def test_library_exports_nothing_but_the_declared_symbols():
    """-fvisibility=hidden + GST_API: the dynamic symbol table's FUNCTIONS are the header's entry points, no C++ internals
    (what remains beside them: weak std:: template instances, and the kernel handle objects hipcc emits per __global__)."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "gstfwd.h")).read()
    declared = set(re.findall(r"\b(gst_[a-z0-9_]+)\s*\(", hdr))
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    funcs, others = set(), []
    for line in out.splitlines():
        parts = line.split()
        if len(parts) != 3:
            continue
        kind, name = parts[1], parts[2]
        if kind in ("T", "t"):
            funcs.add(name)
        elif kind in ("W", "w"):
            others.append(name)
    assert funcs == declared, funcs ^ declared
    assert all(n.startswith(("_ZNSt", "_ZNKSt", "_ZSt", "_ZN9__gnu_cxx", "_ZNK9__gnu_cxx", "_ZTSSt", "_ZTISt", "_ZTVSt", "_ZGVZNKSt", "_ZZNKSt"))
               for n in others), [n for n in others if "St" not in n[:8]][:10]


def test_every_host_copy_goes_through_the_staging_helpers():
    """the device never touches pageable host memory -- every hipMemcpy* with a host side lives in the four helpers of
    gst_abi.cpp (d2h_bytes, h2d_bytes, h2d_async, d2h_rows), the model upload from its own page-locked buffers, and the
    staged row copy of copy_out_dprobs."""
    src_dir = os.path.join(ROOT, "pygsti_amd", "csrc")
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith((".cpp", ".hpp", ".hip")):
            continue
        for i, line in enumerate(open(os.path.join(src_dir, f)).read().splitlines(), 1):
            if not re.search(r"\bhipMemcpy\w*\(", line) or line.lstrip().startswith("//"):
                continue
            if "hipMemcpyDeviceToDevice" in line or "Symbol(" in line:     # (symbol copies: the phase timers of the development builds)
                continue
            assert f == "gst_abi.cpp" and re.search(r"h_stage|h_up|h_model_pinned|\bh, total|mapped|\(dst, d_src|\(d_dst, src|hipMemcpy2DAsync\(dst", line) or \
                (f == "gst_comm.cpp" and ("peer" in line or "h_desc" in line)) or \
                (f == "gst_normal_abi.cpp" and ("G.h_model" in line or "G.h_part" in line)), "%s:%d: %s" % (f, i, line.strip())      # (h_desc: page-locked, 80 bytes)


def _no_gpu():
    return _lib.device_count() == 0


def test_fills_fail_loudly_without_a_device():
    if not _no_gpu():
        pytest.skip("a GPU is present")
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = plan_from_fixture(fx)
    with pytest.raises(_lib.GstDeviceError):
        pl.fill_probs()
    with pytest.raises(_lib.GstDeviceError):
        pl.fill_dprobs()
    with pytest.raises(_lib.GstDeviceError):
        pl.fill_hprobs(idx1=[0], idx2=[1])


def test_call_order_is_checked():
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = _lib.Plan.from_table(fx['D'], 3, 1, 2, fx['nE'], fx['cache_size'], fx['t_dest'], fx['t_start'], fx['t_cache'],
                              fx['t_rho'], fx['row_ptr'], fx['gate_idx'], fx['eff_ptr'], fx['eff_label'], fx['eff_dest'])
    with pytest.raises((_lib.GstError, _lib.GstDeviceError)):
        pl.fill_probs()          # no model yet (or no device): an error either way, never a silent result
    with pytest.raises(ValueError):
        pl.set_param_map([0], [7], [0])       # object index out of range


def test_product_path_never_uses_the_oracle_or_the_reference():
    pkg = os.path.join(ROOT, "pygsti_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                continue
            src = open(os.path.join(dirpath, f)).read()
            assert "/root/reference" not in src, f
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
            assert "liboracle" not in src and "libgst_ref" not in src, f
    # bench.py runs on the GPU box where the reference does not exist (build() may compile oracle/_ref from it
    # in the build container; that is building the checker, not using it)
    assert "/root/reference" not in open(os.path.join(ROOT, "bench.py")).read()


def test_walk_kernels_have_no_fma_outside_division():
    """Bitwise parity needs separate multiply and add.  The only v_fma_f64 allowed in the device code
    are those of the IEEE division expansion (v_div_scale/v_div_fmas/v_div_fixup sequences)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("ROCm llvm tools not found")
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "libgstfwd.so")
        shutil.copy(_lib.LIB_PATH, so)
        subprocess.check_call([objdump, "--offloading", so], stdout=subprocess.DEVNULL)
        cos = glob.glob(so + ".*gfx950*")
        assert cos, "no gfx950 code object embedded in libgstfwd.so"
        # one code object per .hip source; the rule applies to the walk kernels (the analytic and normal-equation
        # kernels are not bit-exact paths and may fuse)
        asms = [subprocess.check_output([objdump, "-d", co]).decode() for co in sorted(cos)]
        asm = "\n".join(t for t in asms if re.search(r"walk_(rows_|base_)?kernel", t))
    n_fma = len(re.findall(r"\bv_fma_f64\b", asm))
    n_div = len(re.findall(r"\bv_div_fixup_f64\b", asm))
    n_mul = len(re.findall(r"\bv_mul_f64\b", asm))
    assert n_mul > 1000
    assert n_div > 0 and n_fma <= 8 * n_div, (n_fma, n_div)


def test_hot_kernels_do_not_use_scratch_memory():
    """A kernel whose arrays (MFMA accumulators, state vectors) end up in scratch memory still computes the right numbers,
    1.5-2x slower -- it happened when an accumulator array was handed to a lambda from inside a run-time loop.  The code
    objects' metadata must show a private segment of 0 bytes for the contraction and the walk kernels."""
    import glob
    import shutil
    import subprocess
    import tempfile
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm llvm tools not found")
    seen = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "libgstfwd.so")
        shutil.copy(_lib.LIB_PATH, so)
        subprocess.check_call([objdump, "--offloading", so], stdout=subprocess.DEVNULL)
        for co in glob.glob(so + ".*gfx950*"):
            notes = subprocess.check_output([readelf, "--notes", co]).decode()
            for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)", notes, re.S):
                seen[m.group(1)] = int(m.group(2))
    hot = {k: v for k, v in seen.items() if re.search(r"analytic_mfma_kernel|walk_kernel|walk_base_kernel|jtj_mfma_lds_kernel", k)}
    assert len(hot) >= 4, sorted(seen)
    # (the FD walk kernels keep a handful of spilled registers, 64-80 bytes per lane, outside their inner loops -- measured,
    #  accepted; an array in scratch shows up as hundreds of bytes)
    bad = {k: v for k, v in hot.items() if "analytic_mfma64" not in k and v > (128 if "walk_kernel" in k else 0)}
    assert not bad, bad


def test_comm_entry_points_fail_loudly_without_a_device():
    """The exchange entry points validate their arguments and need a device; the IPC rendezvous token is plain data."""
    uid = _lib.Comm.unique_id(_lib.TRANSPORT_IPC)
    assert len(uid) == _lib.COMM_ID_BYTES and uid.startswith(b"/gstfwd_")
    assert _lib.Comm.unique_id(_lib.TRANSPORT_IPC) != uid
    with pytest.raises(ValueError):
        _lib.Comm.unique_id(7)                                   # unknown transport
    with pytest.raises(ValueError):
        _lib.Comm(3, 2, uid, 0, _lib.TRANSPORT_IPC)               # rank >= size
    if _no_gpu():
        with pytest.raises(_lib.GstDeviceError):
            _lib.Comm(0, 1, uid, 0, _lib.TRANSPORT_IPC)
        assert _lib.pin_host_array(np.zeros(1 << 20)) is False    # stays pageable, no exception


def test_bad_plan_descriptions_are_rejected_not_fatal():
    """Negative sizes / offsets come back as ValueError (GST_EINVAL): nothing may reach std::terminate across the ABI."""
    fx = load_fixture("smq1Q_XYI_L4_depol")
    args = [fx['D'], 3, 1, 2, fx['nE'], fx['cache_size'], fx['t_dest'], fx['t_start'], fx['t_cache'], fx['t_rho'],
            fx['row_ptr'], fx['gate_idx'], fx['eff_ptr'], fx['eff_label'], fx['eff_dest']]
    bad = list(args); bad[4] = -5
    with pytest.raises(ValueError):
        _lib.Plan.from_table(*bad)
    bad = list(args); ep = np.array(fx['eff_ptr']).copy(); ep[-1] = -1; bad[12] = ep
    with pytest.raises(ValueError):
        _lib.Plan.from_table(*bad)
    n = len(fx['circ_ptr']) - 1
    with pytest.raises(ValueError):
        _lib.Plan.from_circuits(fx['D'], 3, 1, 2, -1, np.zeros(n, np.int32), fx['circ_ptr'], fx['circ_gates'],
                                np.arange(n + 1) * 2, np.tile([0, 1], n), np.arange(2 * n))
