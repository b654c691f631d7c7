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


def _rand_init(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            ln = ("ln_" in n or "LayerNorm" in n or "layer_norm" in n) and n.endswith("weight")
            p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g) if ln else 0.1 * torch.randn(p.shape, generator=g))


@pytest.mark.parametrize("B,L,D,n,layers", [(3, 20, 64, 4, 2), (2, 50, 32, 2, 1)])
def test_gpt2_restatement_matches_hf(B, L, D, n, layers):
    """config exactly as GPT2Config.build passes it (layer_norm_eps is ignored by HF -> 1e-5)."""
    cfg = transformers.GPT2Config(n_embd=D, n_inner=4 * D, n_layer=layers, n_head=n, activation_function="gelu",
                                  initializer_range=0.01, layer_norm_eps=0.03, resid_pdrop=0.0, embd_pdrop=0.0,
                                  attn_pdrop=0.0, n_positions=L, n_ctx=L, vocab_size=1)
    assert cfg.layer_norm_epsilon == 1e-5
    m = transformers.GPT2Model(cfg).eval()
    _rand_init(m, 0)
    x = torch.randn(B, L, D)
    with torch.no_grad():
        ref = m(inputs_embeds=x)[0]
        got = O.gpt2_model(x, O.gpt2_params_from_state(m.state_dict()), n, 1e-5)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,L,D,n,layers", [(3, 20, 64, 4, 2), (2, 33, 32, 2, 1)])
def test_bert_restatement_matches_hf(B, L, D, n, layers):
    """config as BertConfig.build passes it: intermediate_size stays HF's 3072."""
    cfg = transformers.BertConfig(hidden_size=D, num_hidden_layers=layers, num_attention_heads=n, hidden_act="gelu",
                                  initializer_range=0.01, layer_norm_eps=0.03, pad_token_id=0,
                                  max_position_embeddings=L + 2, vocab_size=1,
                                  hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    assert cfg.intermediate_size == 3072
    m = transformers.BertModel(cfg).eval()
    _rand_init(m, 1)
    x = torch.randn(B, L, D)
    with torch.no_grad():
        ref = m(inputs_embeds=x)[0]
        got = O.bert_model(x, O.bert_params_from_state(m.state_dict()), n, 0.03)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,L,D,n,layers", [(4, 20, 64, 4, 2), (3, 9, 32, 2, 1)])
def test_xlnet_padding_mask_restatement_matches_hf(B, L, D, n, layers):
    """the opt-in padding mask (oracle key_padding_mask; kernels: key_len) == HF XLNetModel given an attention_mask:
    padded keys get -1e30 except on the diagonal (modeling_xlnet.py non_tgt_mask)."""
    cfg = transformers.XLNetConfig(
        d_model=D, d_inner=4 * D, n_layer=layers, n_head=n, attn_type="bi", ff_activation="gelu",
        initializer_range=0.01, layer_norm_eps=0.03, dropout=0.0, pad_token_id=0, vocab_size=1, mem_len=1)
    torch.manual_seed(1)
    m = transformers.XLNetModel(cfg).eval()
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(1 + 0.1 * torch.randn_like(p) if "layer_norm.weight" in name else 0.1 * torch.randn_like(p))
    x = torch.randn(B, L, D)
    key_len = torch.tensor([L, 1, L // 2, 3][:B], dtype=torch.int32)
    attn_mask = (torch.arange(L)[None] < key_len[:, None]).float()
    with torch.no_grad():
        ref = m(inputs_embeds=x, attention_mask=attn_mask)[0]
        got = O.xlnet_model(x, [O.xlnet_layer_params_from_hf(l) for l in m.layer], n, 0.03, key_len=key_len)
        unmasked = O.xlnet_model(x, [O.xlnet_layer_params_from_hf(l) for l in m.layer], n, 0.03)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    assert float((got - unmasked).abs().max()) > 1e-3          # the mask matters on these inputs


@pytest.mark.parametrize("arch", ["gpt2", "bert"])
def test_gpt2_bert_padding_mask_restatement_matches_hf(arch):
    """opt-in padding mask of the GPT-2 / BERT bodies == HF given an attention_mask (keys masked for every query).
    Compared on the VALID positions: a fully padded query row is arithmetic noise in both implementations."""
    B, L, D, n = 4, 12, 32, 2
    key_len = torch.tensor([L, 1, 7, 3], dtype=torch.int32)
    attn_mask = (torch.arange(L)[None] < key_len[:, None]).long()
    x = torch.randn(B, L, D)
    if arch == "gpt2":
        cfg = transformers.GPT2Config(n_embd=D, n_inner=4 * D, n_layer=2, n_head=n, activation_function="gelu",
                                      resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, n_positions=L, vocab_size=1)
        m = transformers.GPT2Model(cfg).eval()
        _rand_init(m, 3)
        with torch.no_grad():
            ref = m(inputs_embeds=x, attention_mask=attn_mask)[0]
            got = O.gpt2_model(x, O.gpt2_params_from_state(m.state_dict()), n, 1e-5, key_len=key_len)
    else:
        cfg = transformers.BertConfig(hidden_size=D, num_hidden_layers=2, num_attention_heads=n, hidden_act="gelu",
                                      layer_norm_eps=0.03, max_position_embeddings=L + 2, vocab_size=1,
                                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        m = transformers.BertModel(cfg).eval()
        _rand_init(m, 4)
        with torch.no_grad():
            ref = m(inputs_embeds=x, attention_mask=attn_mask)[0]
            got = O.bert_model(x, O.bert_params_from_state(m.state_dict()), n, 0.03, key_len=key_len)
    valid = attn_mask.bool()
    torch.testing.assert_close(got[valid], ref[valid], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,L,D,n,layers", [(3, 20, 64, 4, 2), (2, 7, 32, 2, 3)])
def test_xlnet_training_mode_restatement_matches_hf(B, L, D, n, layers):
    """The whole-model training-mode composition (oracle xlnet_model_dropout: input dropout, ONE pos_emb mask shared by
    the layers, four sites per layer, output dropout) against the installed HF XLNetModel in .train() with dropout 0.3
    (XLNetConfig.build's default, config/transformer.py:440): the masks HF's nn.Dropout modules drew are recovered by
    forward hooks (kept <=> output != 0 where the input is non-zero; where the input is zero the mask does not matter)
    and handed to the oracle; outputs AND gradients must agree.  This pins the site list, the sharing of the pos_emb
    mask, the 1/(1-p) scaling and the layouts (HF is time-first inside) -- the composition the full-size dropout-0.3
    GPU test (tests/test_round6_gpu.py) then holds the HIP path to."""
    p_drop = 0.3
    cfg = transformers.XLNetConfig(
        d_model=D, d_inner=4 * D, n_layer=layers, n_head=n, attn_type="bi", ff_activation="gelu",
        initializer_range=0.01, layer_norm_eps=0.03, dropout=p_drop, pad_token_id=0, vocab_size=1, mem_len=1)
    torch.manual_seed(2)
    m = transformers.XLNetModel(cfg).train()
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(1 + 0.1 * torch.randn_like(p) if "layer_norm.weight" in name else 0.1 * torch.randn_like(p))
    calls = {}

    def hook(tag):
        def fn(mod, inp, out):
            calls.setdefault(tag, []).append(((out != 0) | (inp[0] == 0)).detach())
        return fn

    hs = [m.dropout.register_forward_hook(hook("model"))]
    for i, l in enumerate(m.layer):
        hs.append(l.rel_attn.dropout.register_forward_hook(hook(f"attn{i}")))
        hs.append(l.ff.dropout.register_forward_hook(hook(f"ff{i}")))
    x = torch.randn(B, L, D, requires_grad=True)
    ref = m(inputs_embeds=x)[0]
    for h in hs:
        h.remove()
    assert len(calls["model"]) == 3 and all(len(calls[f"attn{i}"]) == 2 and len(calls[f"ff{i}"]) == 2 for i in range(layers))
    tf = lambda t: t.transpose(0, 1)            # HF [len, B, ...] -> batch-first
    inp, pos, fin = calls["model"]
    assert inp.shape == (L, B, D) and pos.shape == (2 * L, B, D) and fin.shape == (L, B, D)
    masks = dict(input=tf(inp), pos=tf(pos), final=tf(fin), layers=[])
    for i in range(layers):
        prob, ao = calls[f"attn{i}"]
        act, out = calls[f"ff{i}"]
        assert prob.shape == (B, n, L, L) and act.shape == (L, B, 4 * D)
        masks["layers"].append(dict(prob=prob, attn_out=tf(ao), ff_act=tf(act), ff_out=tf(out)))
    xo = x.detach().clone().requires_grad_()
    lp = [{k: v.detach().clone().requires_grad_() for k, v in O.xlnet_layer_params_from_hf(l).items()} for l in m.layer]
    got = O.xlnet_model_dropout(xo, lp, n, 0.03, masks, p_drop)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)
    w = torch.randn(B, L, D)
    (ref * w).sum().backward()
    (got * w).sum().backward()
    torch.testing.assert_close(xo.grad, x.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(lp[0]["r"].grad, m.layer[0].rel_attn.r.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(lp[-1]["w1"].grad, m.layer[-1].ff.layer_1.weight.grad, rtol=1e-4, atol=2e-5)
    # and the masks mattered
    assert float((got - O.xlnet_model(xo, lp, n, 0.03)).abs().max()) > 1e-2
