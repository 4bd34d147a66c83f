"""Boundary and layout checks that need no GPU: the C-ABI library loads and exports every
function include/slamit.h declares; struct layouts agree between C and the Python binding;
the product never touches oracle/; required files exist."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from tests.helpers import ROOT


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "slamit.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slamit_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from weiner_slamit_v2_amd import api, build

    build.build()
    lib = api.lib()
    declared = _declared_functions()
    assert len(declared) >= 19
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(api.EXPORTS) == declared  # the binding's list is the header's list
    assert b"gfx950" in lib.slamit_version()


def test_no_compute_without_gpu_fails_loudly():
    """On a machine without a HIP device the entry points return SLAMIT_ERR_DEVICE and say why;
    nothing falls back to a CPU path."""
    from weiner_slamit_v2_amd import api

    if api.device_count() > 0:
        return  # on the GPU box this is covered by the parity tests
    h = C.c_void_p()
    p = api.OrbParams(1000, 1.2, 8, 20, 7, 640, 480, 1)
    rc = api.lib().slamit_orb_create(C.byref(p), 0, C.byref(h))
    assert rc == -2 and not h.value and len(api.lib().slamit_last_error()) > 0
    q = np.zeros((4, 32), np.uint8)
    out = np.zeros(4, np.int32)
    rc = api.lib().slamit_hamming_best2(q.ctypes.data, 4, q.ctypes.data, 4, out.ctypes.data, out.ctypes.data, out.ctypes.data)
    assert rc == -2


def test_struct_layouts_match_the_header():
    from weiner_slamit_v2_amd import api

    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "slamit.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(slamit_kp), sizeof(slamit_orb_params), sizeof(slamit_ba_problem),
               sizeof(slamit_ba_opts), sizeof(slamit_ba_stats), sizeof(slamit_ba_result), offsetof(slamit_ba_stats, lambda),
               offsetof(slamit_ba_opts, stop));
        printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(slamit_search_rule), sizeof(slamit_search_batch), sizeof(slamit_frame_view),
               sizeof(slamit_search_queries), sizeof(slamit_camera), sizeof(slamit_bow_groups), sizeof(slamit_bow_rule), offsetof(slamit_bow_rule, kp1_xy),
               sizeof(slamit_sim3_problem), offsetof(slamit_sim3_problem, fix_scale), sizeof(slamit_sim3_result), offsetof(slamit_sim3_result, inlier));
        return 0;
    }'''
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    c = os.path.join(d, "_layout.c")
    open(c, "w").write(src)
    exe = os.path.join(d, "_layout")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])  # the header is plain C
    sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes[0] == 28 == api.KP_DTYPE.itemsize
    assert sizes[1] == C.sizeof(api.OrbParams) and sizes[2] == C.sizeof(api.BaProblem)
    assert sizes[3] == C.sizeof(api.BaOpts) and sizes[4] == C.sizeof(api.BaStats) and sizes[5] == C.sizeof(api.BaResult)
    assert sizes[6] == api.BaStats.lambda_.offset and sizes[7] == api.BaOpts.stop.offset
    want = [C.sizeof(api.SearchRule), C.sizeof(api.SearchBatch), C.sizeof(api.FrameView), C.sizeof(api.SearchQueries), C.sizeof(api.Camera),
            C.sizeof(api.BowGroups), C.sizeof(api.BowRule), api.BowRule.kp1_xy.offset, C.sizeof(api.Sim3Problem), api.Sim3Problem.fix_scale.offset,
            C.sizeof(api.Sim3Result), api.Sim3Result.inlier.offset]
    assert sizes[8:] == want


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under weiner_slamit_v2_amd/ may import, include,
    link or execute it (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "weiner_slamit_v2_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if not f.endswith((".py", ".h", ".hip", ".cc", ".cpp", ".c", "Makefile")):
                continue
            txt = open(os.path.join(dp, f), errors="replace").read()
            if re.search(r"(from|import)\s+oracle\b|oracle/|liborb_oracle|libba_oracle|ba_ref", txt):
                bad.append(os.path.relpath(os.path.join(dp, f), ROOT))
    assert not bad, bad
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", bench)]
    assert uses, "bench.py must time the oracle as its cpu_baseline"
    for u in uses:  # only inside the cpu_baseline* functions
        last_def = max(m.start() for m in re.finditer(r"^def \w+", bench[:u], flags=re.M))
        assert bench[last_def:].startswith("def cpu_baseline"), bench[last_def:last_def + 40]
    ldd = subprocess.check_output(["ldd", os.path.join(pkg, "libslamit_hip.so")]).decode()
    assert "oracle" not in ldd


def test_required_files():
    for f in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/slamit.h",
              "oracle/orb_oracle.cc", "oracle/ba_oracle.cc", "oracle/Makefile", "oracle/Makefile.ref", "profiles"):
        assert os.path.exists(os.path.join(ROOT, f)), f
    gi = open(os.path.join(ROOT, ".gitignore")).read()
    assert "oracle/_ref/" in gi
    assert not os.path.exists(os.path.join(ROOT, ".gpurunignore")) or "oracle/_ref" not in open(os.path.join(ROOT, ".gpurunignore")).read()
