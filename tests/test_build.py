"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so_path():
    from cmflow_amd import _lib
    return _lib.build()


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(REPO, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(REPO, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(cmf_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol(so_path):
    lib = ctypes.CDLL(so_path)
    decl = declared_symbols()
    assert len(decl) >= 20
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.cmf_version is not None
    lib.cmf_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.cmf_version()


def test_ctypes_signatures_cover_the_header(so_path):
    from cmflow_amd import _lib
    decl = declared_symbols() - {"cmf_version"}
    assert decl == set(_lib.SIGNATURES), decl ^ set(_lib.SIGNATURES)


def test_only_gfx950_code_objects(so_path):
    """The fat binary embedded in the library carries gfx950 code objects only (no other arch, no
    compatibility targets)."""
    blob = open(so_path, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert archs == {b"gfx950"}, archs


def test_ball_query_isa_has_no_fma_contraction():
    """The canonical arithmetic of the neighbour search: the ball-query distance must be
    three products and two adds, each individually rounded -- no v_fma/v_fmac/v_mad in that kernel."""
    src = os.path.join(REPO, "cmflow_amd", "csrc", "neighbor.hip")
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                          "--cuda-device-only", "-o", "-", src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    asm = out.stdout
    for sym in ("_Z17ball_query_kernel", "_Z23ball_query_multi_kernelILb0", "_Z23ball_query_multi_kernelILb1",
                "_Z24ball_query_ballot_kernelILi1E", "_Z24ball_query_ballot_kernelILi4E", "_Z22query_and_group_kernelILi1E",
                "_Z22query_and_group_kernelILi2E", "_Z20bq_grid_query_kernel"):
        start = asm.index(sym)
        body = asm[start:asm.index("s_endpgm", start)]
        assert not re.search(r"v_(fma|fmac|mad|pk_fma)_f32", body), sym
    knn = asm[asm.index("_Z10knn_kernelILi8ELb0EE"):]
    knn = knn[:knn.index("s_endpgm")]
    assert re.search(r"v_(fma|fmac)_f32", knn)        # the k-ordered FMA chain of the dot product IS required there
    plain = asm[asm.index("_Z10knn_kernelILi8ELb1EE"):]
    plain = plain[:plain.index("s_endpgm")]
    assert not re.search(r"v_(fma|fmac|mad|pk_fma)_f32", plain)      # extension knn / three_nn: direct form, no FMA


def test_product_fails_loudly_without_gpu():
    """No CPU fallback anywhere on the product path."""
    import torch
    from cmflow_amd.pointnet2_utils import ball_query, grouping_operation
    from cmflow_amd.radarflow_util import knn_point, weighted_kabsch
    x = torch.zeros(1, 8, 3)
    for fn in (lambda: ball_query(1.0, 4, x, x), lambda: knn_point(4, x, x),
               lambda: grouping_operation(torch.zeros(1, 2, 8), torch.zeros(1, 8, 4, dtype=torch.int32)),
               lambda: weighted_kabsch(torch.zeros(1, 3, 8), torch.zeros(1, 3, 8), torch.ones(1, 8) / 8)):
        with pytest.raises(RuntimeError):
            fn()


def test_ctypes_structs_match_the_header(tmp_path):
    """The descriptor structs are filled in from Python: their ctypes mirrors must have the C layout (size and the offset
    of the last member) -- a field appended on one side only would shift every pointer behind it."""
    from cmflow_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cmflow_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(cmf_setconv_desc), offsetof(cmf_setconv_desc, acc_bn),'
                   ' sizeof(cmf_bn_update_entry), offsetof(cmf_bn_update_entry, offset), sizeof(cmf_gemm_launch_record),'
                   ' sizeof(cmf_mlp_desc), offsetof(cmf_mlp_desc, acc_bn)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(_lib.SetConvDesc), _lib.SetConvDesc.acc_bn.offset, ctypes.sizeof(_lib.BnUpdateEntry),
            _lib.BnUpdateEntry.offset.offset, ctypes.sizeof(_lib.GemmLaunchRecord), ctypes.sizeof(_lib.MlpDesc), _lib.MlpDesc.acc_bn.offset]
    assert got == want
