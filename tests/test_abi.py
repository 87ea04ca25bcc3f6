"""The C-ABI library loads and exports every symbol include/vsf.h declares (no compute calls:
this runs without a GPU), and the product path fails loudly when no GPU is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "vsf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vsf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(vsf):
    names = _declared()
    assert len(names) >= 20
    lib = vsf.lib()
    for n in names:
        assert hasattr(lib, n), "libvsf_hip.so does not export %s" % n
    assert sorted(vsf.ABI_SYMBOLS) == names, "python binding list and header disagree"
    assert b"gfx950" in lib.vsf_version()


def test_params_struct_layout(vsf):
    p = vsf.default_params()
    assert (p.grid_rows, p.grid_cols, p.max_features, p.klt_win, p.klt_max_level) == (4, 4, 300, 21, 3)
    assert (p.width, p.height, p.streams) == (752, 480, 1)
    assert p.quality_level == 0.1 and p.min_eig_thr == 1e-4 and p.block_size == 3
    # the header is plain C: compile it with gcc and compare the struct sizes with the ctypes mirror
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "vsf.h"\nint main(void){printf("%zu %zu\\n",'
                             'sizeof(vsf_params),sizeof(vsf_track_stats));printf("%zu %d\\n",sizeof(vsf_info),'
                             'VSF_COMM_ID_BYTES);return 0;}\n')
        exe = os.path.join(d, "sz")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        a, b, c, d = subprocess.check_output([exe]).split()
    assert int(a) == C.sizeof(vsf.VsfParams) and int(b) == C.sizeof(vsf.VsfTrackStats)
    assert int(c) == C.sizeof(vsf.VsfInfo) and int(d) == vsf.COMM_ID_BYTES
    assert p.lk_cache_max_bytes == 0


def test_comm_for_a_world_of_one_needs_no_gpu_and_no_rccl(vsf):
    """vsf_comm_create with nranks == 1 is host-only bookkeeping; bad ranks are refused."""
    lib = vsf.lib()
    h = C.c_void_p()
    assert lib.vsf_comm_create(None, 0, 1, 0, C.byref(h)) == vsf.VSF_OK and h.value
    r, n = C.c_int32(-1), C.c_int32(-1)
    assert lib.vsf_comm_rank(h, C.byref(r), C.byref(n)) == vsf.VSF_OK and (r.value, n.value) == (0, 1)
    lib.vsf_comm_destroy(h)
    for rank, nranks in ((1, 1), (-1, 2), (0, 0)):
        assert lib.vsf_comm_create(None, rank, nranks, 0, C.byref(h)) == vsf.VSF_E_INVALID
    assert lib.vsf_comm_create(None, 0, 2, 0, C.byref(h)) == vsf.VSF_E_INVALID   # several ranks need the unique id


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "vi-slam_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"#include\s*[\"<][^\">]*oracle", r"libvsf_oracle", r"\borc_\w+\s*\("):
                    assert not re.search(pat, txt, flags=re.M), "%s uses the oracle (%s)" % (f, pat)


def test_create_fails_without_gpu(vsf):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vsf.VsfError):
        vsf.Context(vsf.default_params())


def test_bad_parameters_rejected(vsf):
    lib = vsf.lib()
    h = C.c_void_p()
    for kw in ({"klt_win": 40}, {"klt_win": 2}, {"streams": 0}, {"width": 10, "height": 10}, {"block_size": 5},
               {"klt_max_level": 9}, {"grid_rows": 0}, {"quality_level": 1.5}, {"quality_level": -0.1},
               {"width": 65536, "height": 16384},                                  # 32-bit raster indices
               {"detector_kind": 1, "anms_quality_level": 2.0},
               {"detector_kind": 1, "width": 32000, "height": 32000}):              # ANMS tile count above grid.y
        p = vsf.default_params(**kw)
        assert lib.vsf_create(C.byref(p), C.byref(h)) == vsf.VSF_E_INVALID
    assert lib.vsf_create(None, C.byref(h)) == vsf.VSF_E_INVALID
