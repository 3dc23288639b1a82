"""CPU-side checks of the drop-in boundary: the C-ABI shared library builds for gfx950, loads without a GPU, and exports
every symbol include/pixart_hip.h declares; the ctypes binding covers exactly that set.  No compute calls."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "pixart_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pxa_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from pixart_sigma_amd import build, lib
    path = build.build()
    assert os.path.exists(path)
    dll = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/pixart_hip.h but not exported"
    assert set(names) == set(lib.SIGNATURES) | set(lib.OTHER_SYMBOLS)
    assert dll.pxa_abi_version() == lib.ABI_VERSION


def test_gemm_item_hand_out_switch_round_trips():
    """pxa_gemm_set_dynamic_items (ABI 6) is process state, no GPU needed: static split by default, returns the previous setting."""
    import subprocess
    import sys
    code = ("from pixart_sigma_amd import lib; L = lib.load(); a = L.pxa_gemm_set_dynamic_items(1); b = L.pxa_gemm_set_dynamic_items(0); "
            "c = L.pxa_gemm_set_dynamic_items(0); print(a, b, c)")
    env = {k: v for k, v in os.environ.items() if k not in ("PXA_GEMM_STATIC", "PXA_GEMM_DYNAMIC")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[-3:] == ["0", "1", "0"]
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env={**env, "PXA_GEMM_DYNAMIC": "1"})   # an A/B override wins
    assert r.returncode == 0 and r.stdout.split()[-3:] == ["1", "1", "1"]


def test_struct_layouts_match_header():
    """Field order of the ctypes structures follows the C structs (guards against silent ABI drift)."""
    from pixart_sigma_amd import lib
    src = open(os.path.join(ROOT, "include", "pixart_hip.h")).read()
    for cname, st in (("pxa_gemm_args", lib.GemmArgs), ("pxa_attn_args", lib.AttnArgs), ("pxa_grid", lib.GridArg),
                      ("pxa_came_tensor", lib.CameTensor), ("pxa_came_tile", lib.CameTile), ("pxa_came_args", lib.CameArgs)):
        body = re.search(r"typedef struct \{([^}]*)\} " + cname, src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.findall(r"[A-Za-z_0-9]+", part)[-1])
        assert fields == [f[0] for f in st._fields_], cname


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    from pixart_sigma_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(AssertionError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))
