"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header declares, and the
ctypes structs mirror the header's layout.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from titok_video_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "titok_hip.h")


@pytest.fixture(scope="module")
def handle():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.lib()


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ttv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(handle):
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(handle, n), f"{n} declared in titok_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes binding table and header disagree"


def test_version_and_error_string(handle):
    assert handle.ttv_version() >= 100
    assert isinstance(handle.ttv_error_string(), bytes)


def test_struct_layouts_match_header():
    # sizes follow from the field lists in include/titok_hip.h (4-byte ints/floats, 8-byte pointers, natural alignment)
    assert C.sizeof(_lib.FsqParams) == 4 + 6 * 4 * _lib.TTV_MAX_FSQ
    assert C.sizeof(_lib.TowerDims) == 15 * 4
    assert C.sizeof(_lib.LayerWeights) == 25 * 8   # 24 pointers (8 of them the MX fp8 images, round 4) + int32 mlp_pack_qkv_rows + int32 qkv_q_prescaled
    assert C.sizeof(_lib.TowerWeights) == 11 * 8   # 10 pointers + int32 f32_split3 (padded)
    # + blocks64, row_seq, n_blocks64, qblocks_paired, qblocks_all_full (+ pad), items64, n_items64 (+ pad), rope_ids, rope_base,
    # qblocks_latent, n_qblocks_latent (+ pad), qblocks_patch, n_qblocks_patch (+ pad)
    assert C.sizeof(_lib.Batch) == 6 * 4 + 8 * 8 + 16 + 8 + 8 + 16 + 16 + 16
    src = open(HEADER).read()
    for struct, cls in [("ttv_fsq_params", _lib.FsqParams), ("ttv_tower_dims", _lib.TowerDims),
                        ("ttv_layer_weights", _lib.LayerWeights), ("ttv_tower_weights", _lib.TowerWeights),
                        ("ttv_batch", _lib.Batch), ("ttv_layer_weights_t", _lib.LayerWeightsT),
                        ("ttv_tower_weights_t", _lib.TowerWeightsT), ("ttv_layer_grads", _lib.LayerGrads),
                        ("ttv_tower_grads", _lib.TowerGrads)]:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(struct\s+)?[A-Za-z_0-9]+\s*\*?\s*", "", decl, count=1)
            for part in decl.split(","):
                fields.append(re.sub(r"[\*\s]|\[.*\]", "", part))
        assert fields == [f[0] for f in cls._fields_], struct


def test_invalid_arguments_return_codes_without_touching_the_gpu(handle):
    d = _lib.TowerDims(kind=0, dtype=7)
    b = _lib.Batch()
    assert handle.ttv_tower_workspace_bytes(C.byref(d), C.byref(b)) == -1
    assert b"dtype" in handle.ttv_error_string()
    rc = handle.ttv_linear(None, 0, None, 0, None, None, None, 0, 4, 4, 4, 0, None)
    assert rc == 1 and b"null" in handle.ttv_error_string()


def test_workspace_size_is_deterministic(handle):
    d = _lib.TowerDims(kind=0, dtype=0, width=256, layers=4, q_heads=4, kv_heads=2, head_dim=64, inner=704, patch_t=4,
                       patch_h=8, patch_w=8, pix_channels=3, token_size=5, eps=1e-5, alpha=8.0)
    b = _lib.Batch(n_clips=32, total_rows=36864, sum_tokens=4096, sum_patches=32768, max_patches_per_clip=1024, n_qblocks=288)
    n = handle.ttv_tower_workspace_bytes(C.byref(d), C.byref(b))
    L, P = 36864, 32768
    expect = sum(((x + 255) // 256) * 256 for x in
                 [L * 256 * 2, L * 256 * 2, L * 768 * 2, L * 256 * 2, L * 256 * 4, L * 704 * 2, P * 768 * 2, P * 256 * 2, L * 4,
                  4096 * 256 * 2, 4096 * 256 * 2])   # + rstd; + the compact latent rows of x and of the attention output (encoder, last layer)
    assert n == expect


def test_hip_adamw_has_no_cpu_path():
    """optim.HipAdamW steps on the GPU only: parameters on the host raise, and make_optimizer() gives host parameters torch's AdamW."""
    import pytest
    import torch
    from titok_video_amd.optim import HipAdamW
    from titok_video_amd.train import make_optimizer
    p = torch.nn.Parameter(torch.ones(8))
    p.grad = torch.ones(8)
    opt = HipAdamW([p], lr=1e-3)
    with pytest.raises(RuntimeError, match="GPU only"):
        opt.step()
    lin = torch.nn.Linear(4, 4)
    assert type(make_optimizer(lin)) is torch.optim.AdamW
