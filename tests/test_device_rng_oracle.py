"""CPU: the oracle's restatement of the device random streams (oracle/device_rng.py + device_rng.c).
Pinned against the published Philox4x32-10 known-answer vectors (Random123 kat_vectors); numpy form == C form; the
derived rules (threshold, counter layout, MLM selectors) at their edge cases.  The comparison with masks exported from
the device is tests/test_round6_gpu.py (-m gpu)."""
import ctypes

import numpy as np
import pytest
import torch

import device_rng as R
import t4r_oracle as O

KAT = [  # counter[4], key[2], expected[4]  (Random123 kat_vectors: philox4x32 10)
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.fixture(scope="module", autouse=True)
def _c_form():
    import build_c

    build_c.build()
    assert R.use_c()


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(ctr, key, want):
    seed, lo, hi = key[0] | (key[1] << 32), ctr[0] | (ctr[1] << 32), ctr[2] | (ctr[3] << 32)
    got = R.philox4x32_10(seed, np.array([lo], dtype=np.uint64), hi)[0]
    assert tuple(int(x) for x in got) == want
    out = (ctypes.c_uint32 * 4)()
    R._load_c().t4r_oracle_philox(seed, lo, hi, out)
    assert tuple(out) == want


def test_numpy_and_c_forms_agree():
    for n in (1, 2, 3, 4, 5, 1023, 100_003):
        for p in (0.1, 0.3, 0.5):
            ctr = R.dropout_ctr_hi(7, 3, R.SITE_FF_ACT)
            assert np.array_equal(R.dropout_keep(2**63 - 5, ctr, n, p), R.dropout_keep(2**63 - 5, ctr, n, p, force_numpy=True))
    for B, L in ((1, 1), (7, 20), (64, 33)):
        a, b = R.mlm_draws(99, 12345, B, L, 0.15), R.mlm_draws(99, 12345, B, L, 0.15, force_numpy=True)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_threshold_and_counter_layout():
    assert R.keep_threshold(0.0) == 0 and R.keep_threshold(1.0) == 0xFFFFFFFF
    # float32(0.3) * 2^24 = 5033165 exactly (0.3f = 10066330 * 2^-25): keep <=> top 24 bits >= 5033165
    assert R.keep_threshold(0.3) == 5033165 << 8
    assert R.keep_threshold(0.5) == 1 << 31
    assert R.dropout_ctr_hi(1, 255, R.SITE_POS) == (1 << 16) | (255 << 8) | 1
    assert R.dropout_ctr_hi(3, 2, R.SITE_FF_OUT) == (3 << 16) | (2 << 8) | 5
    # prefix property: element idx does not depend on how many elements the site has
    a, b = R.dropout_keep(5, 9, 1000, 0.3), R.dropout_keep(5, 9, 37, 0.3)
    assert np.array_equal(a[:37], b)
    keep = R.dropout_keep(1, R.dropout_ctr_hi(1, 0, 2), 1 << 20, 0.3)
    assert abs(keep.mean() - 0.7) < 2e-3
    # sites / layers / forwards are different streams
    streams = [R.dropout_keep(1, R.dropout_ctr_hi(o, l, s), 4096, 0.3) for o, l, s in ((1, 0, 2), (2, 0, 2), (1, 1, 2), (1, 0, 3))]
    assert all(not np.array_equal(streams[0], s) for s in streams[1:])


def test_mlm_targets_from_device_draws_follow_the_reference_rule():
    """the selector arithmetic (k-th non-pad / labelled position) gives what t4r_oracle.mlm_targets_train gives for the
    same (bern, j1, j2) -- the function the reference fixtures pin -- plus the reference's invariants
    (tests/unit/torch/test_masking.py:117-150): every session with >= 2 items has >= 1 label and < all its items."""
    g = torch.Generator().manual_seed(0)
    B, L = 257, 20
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, 1000, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    mask, labels = R.mlm_targets_train_device(ids, seed=4321, offset=3 * B * L, p=0.15)
    bern, u1, u2 = R.mlm_draws(4321, 3 * B * L, B, L, 0.15)
    nonpad = ids != 0
    n_np = nonpad.sum(1)
    k1 = torch.minimum(n_np - 1, (torch.from_numpy(u1) * n_np.float()).long())
    j1 = torch.stack([torch.nonzero(nonpad[b])[k1[b], 0] for b in range(B)])

    def j2_fn(m):
        n = m.sum(1)
        k = torch.minimum(n - 1, (torch.from_numpy(u2) * n.float()).long()).clamp_min(0)
        return torch.stack([torch.nonzero(m[b])[k[b], 0] if n[b] > 0 else torch.tensor(0) for b in range(B)])

    m_ref, lab_ref = O.mlm_targets_train(ids, torch.from_numpy(bern).bool(), j1, j2_fn)
    assert torch.equal(mask, m_ref) and torch.equal(labels, lab_ref)
    multi = lens >= 2
    assert bool((mask.sum(1)[multi] >= 1).all()) and bool((mask.sum(1)[multi] < lens[multi]).all())
    assert bool((mask.sum(1)[~multi] == 0).all())            # a one-item session ends with no label (masking.py:447-457)
    assert torch.equal(labels, torch.where(mask, ids, torch.zeros_like(ids)))


def test_mask_dictionary_shapes():
    m = R.xlnet_dropout_masks(3, 5, 8, 2, 2, 0.3, seed=1, offset=1)
    assert m["input"].shape == (3, 5, 8) and m["pos"].shape == (3, 10, 8) and m["final"].shape == (3, 5, 8)
    assert len(m["layers"]) == 2 and m["layers"][1]["prob"].shape == (3, 2, 5, 5) and m["layers"][0]["ff_act"].shape == (3, 5, 32)
    assert not torch.equal(m["layers"][0]["attn_out"], m["layers"][1]["attn_out"])
