"""The XLNet block arithmetic lives in a third-party dependency of the reference
(HuggingFace `transformers`, pinned >=4.12,<4.31.0 in requirements/base_external.txt:1;
installed here: 5.x).  This pins the oracle's batch-first restatement against the installed
HF XLNetModel live, with the hyper-parameters XLNetConfig.build sets
(transformers4rec/config/transformer.py:432-482).  CPU only."""
import pytest
import torch

import t4r_oracle as O

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("B,L,D,n,layers", [(3, 20, 64, 4, 2), (2, 21, 32, 2, 1), (2, 7, 128, 4, 1)])
def test_xlnet_restatement_matches_hf(B, L, D, n, layers):
    cfg = transformers.XLNetConfig(
        d_model=D, d_inner=4 * D, n_layer=layers, n_head=n, attn_type="bi", ff_activation="gelu",
        initializer_range=0.01, layer_norm_eps=0.03, dropout=0.0, pad_token_id=0, vocab_size=1,
        mem_len=1)
    torch.manual_seed(0)
    m = transformers.XLNetModel(cfg).eval()
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(1 + 0.1 * torch.randn_like(p) if "layer_norm.weight" in name else 0.1 * torch.randn_like(p))
    x = torch.randn(B, L, D)
    with torch.no_grad():
        ref = m(inputs_embeds=x)[0]
        got = O.xlnet_model(x, [O.xlnet_layer_params_from_hf(l) for l in m.layer], n, 0.03)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


def test_rel_shift_identity():
    """rel_shift_bnij (HF modeling_xlnet.py:81-93) == gather at j + L - i for klen == qlen."""
    L = 9
    x = torch.randn(2, 3, L, 2 * L)
    ref = transformers.models.xlnet.modeling_xlnet.XLNetRelativeAttention.rel_shift_bnij(x, klen=L)
    idx = torch.arange(L)[None, :] + L - torch.arange(L)[:, None]
    got = torch.gather(x, 3, idx[None, None].expand(2, 3, L, L))
    assert torch.equal(ref, got)
