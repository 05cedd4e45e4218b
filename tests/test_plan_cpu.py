"""Host logic: batch plan tables (cu_seqlens, row maps, q-blocks, RoPE table) against the oracle / reference fixtures;
module state-dict compatibility with the reference's key names.  CPU only."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import titok_oracle as O
from titok_video_amd.plan import BatchPlan, get_plan, host_ints
from titok_video_amd.model.titok import TiTok
from titok_video_amd.model.quantizer.fsq import FSQ
from titok_video_amd.synthetic import seeded_titok_state, tower_param_shapes

G = os.path.join(os.path.dirname(__file__), "golden")


def test_plan_tables_match_oracle_metadata():
    shapes = [(8, 32, 48), (4, 8, 24), (16, 128, 128), (4, 8, 8)]
    counts = [5, 0, 128, 1]
    plan = BatchPlan(shapes, counts, (4, 8, 8), "cpu")
    grids, sizes, cts, cu, mask = O.batch_metadata(shapes, counts, (4, 8, 8))
    assert plan.cu_seqlens == cu and plan.grid_sizes == sizes
    assert plan.cu_dev.tolist() == cu
    rows = torch.arange(len(mask))
    assert plan.latent_rows_dev.tolist() == rows[mask].tolist()
    assert plan.patch_rows_dev.tolist() == rows[~mask].tolist()
    for hq, hkv in [(4, 2), (12, 4)]:
        for split in (None, False, True):
            tab = plan.attention_table(hq, hkv, split).tolist()
            covered = []
            unit_xcd = {}
            seen_half = [False] * 8
            for i, (b, q0, head, mode) in enumerate(tab):
                if b < 0:
                    continue                                   # padding entry of the XCD-interleaved order
                s_len = cu[b + 1] - cu[b]
                rows = 64 if mode else 128
                assert mode in (0, 1) and q0 % rows == 0 and q0 < s_len and 0 <= head < hq
                assert mode or not seen_half[i % 8]            # half items come after the full ones of their XCD list
                seen_half[i % 8] |= bool(mode)
                assert split is not False or mode == 0
                assert split is not True or mode == 1
                covered += [(b, head, r) for r in range(q0, min(q0 + rows, s_len))]
                unit = (b, head // (hq // hkv))
                assert unit_xcd.setdefault(unit, i % 8) == i % 8    # all blocks sharing K/V sit on one XCD slot
            assert len(covered) == cu[-1] * hq == len(set(covered))   # every (row, q-head) exactly once
        assert plan.batch_for(hq, hkv).n_qblocks == len(plan.attention_table(hq, hkv))


def test_attention_table_half_items_for_small_grids_only():
    """Fewer items than resident slots (1024): blocks 6..8 of every 9-block sequence become half items, chosen per sequence;
    the benchmark batch (1152 items) keeps full items."""
    plan = BatchPlan([(16, 128, 128)] * 8, [128] * 8, (4, 8, 8), "cpu")
    tab = plan.attention_table(4, 2)
    real = tab[tab[:, 0] >= 0]
    assert int((real[:, 3] == 0).sum()) == 8 * 6 * 4 and int((real[:, 3] == 1).sum()) == 8 * 3 * 4 * 2
    one = BatchPlan([(16, 128, 128)], [128], (4, 8, 8), "cpu").attention_table(4, 2)
    one = one[one[:, 0] >= 0]
    mine = real[real[:, 0] == 7][:, 1:]
    assert sorted(map(tuple, mine.tolist())) == sorted(map(tuple, one[:, 1:].tolist()))   # same items alone and in the batch
    big_plan = BatchPlan([(16, 128, 128)] * 32, [128] * 32, (4, 8, 8), "cpu")
    big = big_plan.attention_table(4, 2)
    assert int((big[:, 3] == 1).sum()) == 0 and int((big[:, 0] >= 0).sum()) == 1152
    # ttv_batch.qblocks_all_full (attention flag TTV_ATTN_ALLFULL) follows the table actually handed to the library
    assert big_plan.batch_for(4, 2).qblocks_all_full == 1
    assert plan.batch_for(4, 2).qblocks_all_full == 0


def test_plan_rope_table_is_reference_bits():
    d = np.load(os.path.join(G, "rope_kat.npz"))
    for i in range(4):
        grids, counts = d[f"grids_{i}"].tolist(), d[f"counts_{i}"].tolist()
        plan = BatchPlan([(g[0] * 4, g[1] * 8, g[2] * 8) for g in grids], counts, (4, 8, 8), "cpu")
        cs = plan.rope_cs.numpy()
        assert np.array_equal(cs[:, :30], d[f"cos_{i}"].astype(np.float32))
        assert np.array_equal(cs[:, 32:62], d[f"sin_{i}"].astype(np.float32))
        assert np.all(cs[:, 30:32] == 1.0) and np.all(cs[:, 62:64] == 0.0)


def test_plan_rejects_bad_shapes_and_caches():
    with pytest.raises(ValueError):
        BatchPlan([(5, 8, 8)], [1], (4, 8, 8), "cpu")
    with pytest.raises(ValueError):
        BatchPlan([(4, 8, 8)], [1, 2], (4, 8, 8), "cpu")
    a = get_plan([(4, 8, 8)], [3], (4, 8, 8), "cpu")
    b = get_plan([[4, 8, 8]], torch.tensor([3]).tolist(), [4, 8, 8], "cpu")
    assert a is b
    assert host_ints(torch.tensor([[4, 8, 8]], dtype=torch.int32)) == [[4, 8, 8]]


def test_state_dict_keys_and_param_count_match_reference():
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(
        patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5], encoder_size="tiny", decoder_size="tiny")))
    m = TiTok(cfg)
    keys = list(m.state_dict().keys())
    expect = ["encoder." + k for k in tower_param_shapes("encoder", "tiny", (4, 8, 8), 3, 5)] + \
             ["decoder." + k for k in tower_param_shapes("decoder", "tiny", (4, 8, 8), 5, 3)]
    assert sorted(keys) == sorted(expect)                       # FSQ contributes no keys (non-persistent buffers)
    assert sum(p.numel() for p in m.parameters()) == 6828295
    m.load_state_dict(seeded_titok_state(0), strict=True)
    # reference init (utils.py:54-66): gains 1, biases 0, |w| <= 2 std
    m2 = TiTok(cfg)
    with torch.no_grad():
        assert float(m2.encoder.ln_post.weight.min()) == 1.0 and float(m2.decoder.proj_out.bias.abs().max()) == 0.0
        assert abs(float(m2.encoder.model_layers.attn_layer[0].to_qkv.weight.std()) - 0.02) < 2e-3


def test_fsq_module_constants_match_reference_module_api():
    d = np.load(os.path.join(G, "fsq_kat.npz"))
    f = FSQ([7, 5, 5, 5, 5])
    assert f.codebook_size == 4375 and f.codebook_dim == 5 and f.dim == 5
    assert f._basis.tolist() == [1, 7, 35, 175, 875]
    assert np.array_equal(f.implicit_codebook.numpy(), d["codebook_a"])
    z = torch.from_numpy(d["z_a"])
    assert np.array_equal(f.bound(z).numpy(), d["bounded_a"])
    assert np.array_equal(f.codes_to_indices(torch.from_numpy(d["codes_a"])).numpy(), d["indices_a"])
    assert list(f.state_dict().keys()) == []
    with pytest.raises(RuntimeError):
        f(z)          # CPU tensor: the hot path has no fallback


def test_bench_flop_accounting_matches_the_survey():
    """bench.py's algorithmic FLOP figures: 26.475 GFLOP per clip for tiny/tiny at K = 128, attention
    4 S^2 d per layer-tower launch over the batch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    S, P, K = 1152, 1024, 128
    assert abs(bench.tower_flops_per_clip(S, P, K) - 2.6475e10) / 2.6475e10 < 1e-3
    assert bench.kernel_flops_per_launch("attention", 32, S) == 32 * 4 * S * S * 256


def test_forward_pipeline_needs_a_gpu_model():
    from types import SimpleNamespace
    from titok_video_amd.model.titok import TiTok
    from titok_video_amd.pipeline import ForwardPipeline
    cfg = SimpleNamespace(tokenizer=SimpleNamespace(model=SimpleNamespace(patch_size=[4, 8, 8], fsq_levels=[7, 5, 5, 5, 5],
                                                                          encoder_size="tiny", decoder_size="tiny")))
    with pytest.raises(RuntimeError):
        ForwardPipeline(TiTok(cfg), depth=2)
    with pytest.raises(ValueError):
        ForwardPipeline(TiTok(cfg), depth=0)


def test_attention_table64_covers_every_row_slice_once():
    """Work table of ttv_attention64: every (sequence, q-head, 64-row slice) appears in exactly one wave slot, all four waves of a
    workgroup share the (sequence, kv-head), and the entries i, i + 8, ... of one list keep a unit together."""
    from titok_video_amd.plan import BatchPlan
    plan = BatchPlan([(8, 32, 48), (4, 8, 24), (16, 128, 128), (4, 16, 16)], [5, 3, 128, 1], (4, 8, 8), "cpu")
    for hq, hkv in ((4, 2), (12, 4), (2, 2)):
        t = plan.attention_table64(hq, hkv).numpy()
        rep = hq // hkv
        seen = set()
        for e in t:
            if e[0] < 0:
                continue
            s = plan.cu_seqlens[e[0] + 1] - plan.cu_seqlens[e[0]]
            for w in e[2:6]:
                if w < 0:
                    continue
                head, q64 = int(w) & 0xff, int(w) >> 8
                assert head // rep == e[1] and q64 * 64 < s
                assert (int(e[0]), head, q64) not in seen
                seen.add((int(e[0]), head, q64))
        want = {(b, h, q) for b in range(4) for h in range(hq) for q in range(-(-(plan.cu_seqlens[b + 1] - plan.cu_seqlens[b]) // 64))}
        assert seen == want
    # the benchmark batch: 32 x 1152 rows, 4/2 heads -> 576 workgroups, no idle wave
    big = BatchPlan([(16, 128, 128)] * 32, [128] * 32, (4, 8, 8), "cpu").attention_table64(4, 2).numpy()
    assert big.shape[0] == 576 and (big[:, 2:6] >= 0).all()


def test_attention_backward_blocks_are_a_permutation_grouped_by_xcd():
    """plan._xcd_interleave: every (sequence, first row) block appears exactly once; for a uniform batch every sequence's blocks sit in
    one residue class of the table index modulo 8 (one XCD under round-robin dispatch); ragged batches keep the permutation property."""
    from titok_video_amd.plan import _xcd_interleave
    units = [[(b, r) for r in range(0, 1152, 64)] for b in range(32)]
    t = _xcd_interleave(units)
    assert sorted(map(tuple, t.tolist())) == sorted(e for u in units for e in u)
    for b in range(32):
        assert len({i % 8 for i in range(len(t)) if t[i, 0] == b}) == 1
    ragged = [[(b, r) for r in range(0, n, 64)] for b, n in enumerate([100, 1152, 64, 700, 33, 2000, 65, 640, 9, 1281])]
    t = _xcd_interleave(ragged)
    assert sorted(map(tuple, t.tolist())) == sorted(e for u in ragged for e in u)
