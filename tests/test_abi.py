"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/aligner_amd.h declares; the product never touches the oracle
and fails loudly without a GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "aligner_amd.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aligner_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(built_lib):
    from aligner_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 12
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/aligner_amd.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"


def test_abi_version_and_errors(built_lib):
    assert built_lib.aligner_abi_version() == 4
    assert built_lib.aligner_maxpath_workspace_bytes(64, 200, 1000) > 64 * 32 * 256 * 4
    assert built_lib.aligner_maxpath_workspace_bytes(1, 0, 5) == 0
    # argument validation happens before any HIP call
    rc = built_lib.aligner_maxpath_expand(None, None, 0, 1, 1, 1, None)
    assert rc == -22 and b"null" in built_lib.aligner_last_error()
    rc = built_lib.aligner_softattn_f32(1, 1, None, None, 1, None, 1, 1 << 30, 1, 300, 4, 4, 1.0, 0, None)
    assert rc == -33


def test_product_fails_loudly_without_gpu(built_lib):
    if built_lib.aligner_device_count() > 0:
        pytest.skip("a GPU is visible")
    import aligner_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        aligner_amd.maximum_path(torch.zeros(1, 2, 3), torch.ones(1, 2, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        aligner_amd.maximum_path_c(np.zeros((1, 2, 3), np.int32), np.zeros((1, 2, 3), np.float32),
                                   np.array([2], np.int32), np.array([3], np.int32))


def test_wrapper_argument_checks_mirror_the_reference(built_lib):
    """Same rejections as the Cython glue (core.c:27882-27913, 27843), raised before any GPU work."""
    import aligner_amd
    v = torch.zeros(2, 3, 4)
    with pytest.raises(ValueError, match="not C-contiguous"):
        aligner_amd.maximum_path(v.transpose(1, 2), torch.ones(2, 4, 3))
    with pytest.raises(ValueError, match="dimensions"):
        aligner_amd.maximum_path(v[0], torch.ones(3, 4))
    p = np.zeros((1, 2, 3), np.int32)
    val = np.zeros((1, 2, 3), np.float32)
    t = np.array([2], np.int32)
    with pytest.raises(ValueError, match="dtype mismatch"):
        aligner_amd.maximum_path_c(p, val.astype(np.float64), t, t)
    with pytest.raises(ValueError, match="dimensions"):
        aligner_amd.maximum_path_c(p[0], val[0], t, t)
    ro = val.copy()
    ro.setflags(write=False)
    with pytest.raises(ValueError, match="read-only"):
        aligner_amd.maximum_path_c(p, ro, t, t)


def test_dropin_import_paths(built_lib):
    import aligner_amd
    aligner_amd.install_dropin()
    from monotonic_align import maximum_path                       # reference __init__.py:6
    from monotonic_align.monotonic_align.core import maximum_path_c  # reference __init__.py:3
    assert maximum_path is aligner_amd.maximum_path
    assert maximum_path_c is aligner_amd.maximum_path_c


def test_product_never_imports_the_oracle():
    """aligner_amd/, bench.py's product legs and the C sources must not reference oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "aligner_amd")):
        if os.sep + "lib" in base:
            continue
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                with open(os.path.join(base, fn)) as f:
                    txt = f.read()
                if re.search(r"\boracle\b", txt) and fn != "__init__.py":
                    # the word may only appear in comments that say the oracle is NOT used
                    for line in txt.splitlines():
                        if re.search(r"^\s*(from|import)\s+.*oracle", line) or "maxpath_oracle" in line \
                                or "libmaxpath_oracle" in line:
                            bad.append((fn, line.strip()))
    # bench.py: the oracle may only be imported inside cpu_baseline() (the reported CPU leg, after the timed region)
    import ast
    with open(os.path.join(ROOT, "bench.py")) as f:
        src = f.read()
    tree = ast.parse(src)
    allowed = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("cpu_baseline", "_cpu_dp_baseline"):
            allowed.update(id(n) for n in ast.walk(node))
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)) and id(node) not in allowed:
            names = [a.name for a in node.names] + [getattr(node, "module", None) or ""]
            if any("oracle" in n for n in names):
                bad.append(("bench.py", ast.get_source_segment(src, node)))
        if isinstance(node, ast.Constant) and isinstance(node.value, str) and id(node) not in allowed:
            if "libmaxpath_oracle" in node.value or "oracle/_ref" in node.value:
                bad.append(("bench.py", node.value[:60]))
    assert not bad, bad


@pytest.mark.parametrize("name", ["gen_sweep_asm", "gen_walk_asm"])
def test_generated_asm_is_in_sync(tmp_path, name):
    """aligner_amd/csrc/maxpath_{sweep,walk}_asm.inc are generated (tools/gen_*_asm.py) and committed: the
    committed text must be what the generator writes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    gen.OUT = str(tmp_path / "generated.inc")
    gen.main()
    assert open(gen.OUT).read() == committed


def test_no_kernel_spills_or_uses_scratch():
    """The compiler's resource report of every kernel in the library (kept beside the objects by the build:
    aligner_amd/lib/obj/*.resources.txt, `-Rpass-analysis=kernel-resource-usage`): no spilled VGPR, no scratch.
    Besides the speed: the kernels that issue their loads by hand (inline-asm global_load / LDS-DMA with counted
    s_waitcnt) are only correct while the compiler never moves, spills or reuses a destination register behind an
    in-flight load -- a kernel that spills gives no such guarantee (rules 1 and 11).  The two exceptions are
    named: neither issues a load by hand."""
    import glob
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    files = glob.glob(os.path.join(kr.OBJ, "*.hip.resources.txt"))
    assert len(files) >= 6, "no resource reports: build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    rows = [(os.path.basename(f), name, v) for f in sorted(files) for name, v in kr.parse(f)]      # (mangled names)
    assert len(rows) > 300
    allowed = ("fwdsum_backward_sys_kernelILi4ELi16ELb0E", "fwdsum_backward_sys_kernelILi8ELi8ELb0E",   # compiler-scheduled stager
               "mobo_norm_kernel")                                                                       # (tensors not 16-byte aligned)
    bad = [(src, name, v.get("VGPRs Spill", 0), v.get("ScratchSize [bytes/lane]", 0)) for src, name, v in rows
           if (v.get("VGPRs Spill", 0) or v.get("ScratchSize [bytes/lane]", 0)) and not any(a in name for a in allowed)]
    assert not bad, bad
    for key in ("softattn_rt_kernel", "maxpath_pipelined_kernel", "expand_kernel", "conv_gemm_kernel", "fwdsum_both_sys_kernel",
                "fwdsum_backward_sys_kernelILi4ELi16ELb1E"):
        assert any(key in name for _, name, _ in rows), key


def test_row_pitch_helpers_and_debug_options_env(built_lib, monkeypatch):
    """Host logic of the pipeline's row pitch (no GPU): pitched_logp() gives rows on whole 128-byte lines and never touches
    the reference's layouts; the *_ld entry points refuse a pitch below T_mel and a mask before they look for a device;
    ALIGNER_DEBUG_OPTIONS rejects a name the library does not know."""
    import torch
    from aligner_amd import _lib
    from aligner_amd.softattn import pitched_logp
    for (B, Tx, Ty, dt, ld) in [(2, 200, 1000, torch.float32, 1024), (3, 5, 1024, torch.float32, 1024), (2, 7, 4000, torch.bfloat16, 4032),
                                (1, 1, 36, torch.float32, 64)]:
        t = pitched_logp(B, Tx, Ty, "cpu", dt)
        assert tuple(t.shape) == (B, Tx, Ty) and t.stride(2) == 1 and t.stride(1) == ld and t.stride(0) == Tx * ld
        assert (ld * t.element_size()) % 128 == 0 and (B * Tx == 1 or t.is_contiguous() == (ld == Ty))
    lib = _lib.load()
    buf = torch.zeros(64, dtype=torch.float32)
    p = buf.data_ptr()
    rc = lib.aligner_softattn_ld(p, p, None, None, p, _lib.DT_F32, 3, None, p, 0, 1, 80, 4, 4, 0.0005, _lib.SIM_L2, None)
    assert rc == _lib.EINVAL and b"ld_logp" in lib.aligner_last_error()
    rc = lib.aligner_maxpath_ld(p, _lib.DT_F32, 3, p, p, None, 0, None, None, p, 1 << 20, 1, 4, 4, -1e9, 0, None)
    assert rc == _lib.EINVAL and b"ld_value" in lib.aligner_last_error()
    rc = lib.aligner_maxpath_ld(p, _lib.DT_F32, 8, p, p, None, 0, None, None, p, 1 << 20, 1, 4, 4, -1e9, _lib.F_STRICT_MASK, None)
    assert rc == _lib.EINVAL
    assert lib.aligner_debug_set_option(b"no_such_option", 1) != 0
    assert lib.aligner_debug_set_option(b"conv_no_ring", 0) == 0
