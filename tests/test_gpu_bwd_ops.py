"""Operator-level numerics of the training path's C-ABI entries against plain fp32 torch on the CPU (autograd of the same op
written with torch primitives - nn.LSTM with packed sequences, softmax attention with the box bias, Linear + ReLU), on shapes
the end-to-end goldens do not visit: odd row / column counts (the element-wise GEMM path), uneven heads, one-row batches, long K
(split-K), few rows against wide weights (the weight-stream kernel), sentences of every length incl. 1. Tolerance 2e-5 of each
tensor's largest entry (fp32 both sides, different summation orders)."""
import importlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
bwd = importlib.import_module("vognet-pytorch_amd.backward")
TOL = 2e-5


def close(a, b, tol=TOL, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-12)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, (what, err)
    return err


@pytest.mark.parametrize("M,N,K,rep,relu", [(64, 32, 48, 1, True), (7, 5, 9, 1, True), (33, 257, 130, 1, False), (12, 16, 4096, 1, True),
                                           (4, 4096, 1024, 1, False), (16, 256, 2048, 1, True), (20, 24, 36, 5, True), (1, 3, 2, 1, True),
                                           (4000, 64, 64, 1, True)])
def test_linear_f32_forward_backward(M, N, K, rep, relu):
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    wide = N + 6
    dy = torch.randn(M * rep, wide, generator=g)                       # the layer's output gradient sits inside a wider matrix
    col0 = 3
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = xr @ wr.t() + br
    y = torch.relu(y) if relu else y
    yrep = y.unsqueeze(1).expand(M, rep, N).reshape(M * rep, N)          # rows replicated downstream
    (yrep * dy[:, col0:col0 + N]).sum().backward()
    r = bwd.linear_f32(x.cuda(), w.cuda(), b.cuda(), relu, dy=dy.cuda().contiguous(), dy_col0=col0, rep=rep, want_dx=True, want_y=True)
    torch.cuda.synchronize()
    close(r["y"], y, what="y"); close(r["g_w"], wr.grad, what="g_w"); close(r["g_b"], br.grad, what="g_b"); close(r["d_x"], xr.grad, what="d_x")


def _attn_ref(x, wq, wk, wv, boxes, pe_w, pe_b, n_heads, nsrl, drop_p=0.0):
    S, N, d = x.shape
    q, k, v = x @ wq.t(), x @ wk.t(), x @ wv.t()
    c = -(-d // n_heads)
    outs, off = [], 0
    for h in range(n_heads):
        dh = min(c, d - off)
        lg = q[..., off:off + dh] @ k[..., off:off + dh].transpose(1, 2)
        if boxes is not None:
            diff = boxes.unsqueeze(2) - boxes.unsqueeze(1)
            bh = torch.relu(diff @ pe_w[h] + pe_b[h])
            lg = lg + bh.repeat(1, nsrl, nsrl)
        p = torch.softmax(lg / math.sqrt(d), dim=-1)
        outs.append(p @ v[..., off:off + dh])
        off += dh
    return torch.cat(outs, -1)


@pytest.mark.parametrize("S,n,nsrl,d,H,rel", [(3, 7, 1, 32, 3, True), (2, 5, 3, 48, 3, True), (1, 9, 2, 20, 4, False), (5, 33, 1, 64, 8, True),
                                             (2, 1, 1, 8, 2, True), (3, 100, 1, 512, 3, True)])
def test_attention_f32_forward_backward(S, n, nsrl, d, H, rel):
    g = torch.Generator().manual_seed(S * 1000 + n * 10 + d)
    N = n * nsrl
    x = torch.randn(S, N, d, generator=g)
    ws = [torch.randn(d, d, generator=g) / math.sqrt(d) for _ in range(3)]
    props = torch.rand(S * n, 7, generator=g) * torch.tensor([720., 405., 720., 405., 10., 1., 1.])
    vw, vh, fdiv = 720.0, 405.0, 10.0
    pe_w, pe_b = torch.randn(H, 5, generator=g), torch.randn(H, generator=g) * 0.3
    d_cat = torch.randn(S, N, d, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in [x] + ws + [pe_w, pe_b]]
    bx = None
    if rel:
        bx = (props[:, :5] / torch.tensor([vw, vh, vw, vh, fdiv])).reshape(S, n, 5)
    cat = _attn_ref(leaves[0], leaves[1], leaves[2], leaves[3], bx, leaves[4], leaves[5], H, nsrl)
    (cat * d_cat).sum().backward()
    w = {"wq": ws[0].cuda(), "wk": ws[1].cuda(), "wv": ws[2].cuda()}
    boxes = bwd._Boxes(props.cuda(), vw, vh, fdiv) if rel else None
    pe = (pe_w.cuda(), pe_b.cuda()) if rel else None
    xd = x.reshape(S * N, d).cuda().contiguous()
    f = bwd._attn_call(w, pe, xd, S, N, n, H, boxes)
    close(f["cat"], cat.reshape(S * N, d), what="cat")
    r = bwd._attn_call(w, pe, xd, S, N, n, H, boxes, d_cat=d_cat.reshape(S * N, d).cuda().contiguous())
    torch.cuda.synchronize()
    close(r["d_x"], leaves[0].grad.reshape(S * N, d), what="d_x")
    for i, k in enumerate(("wq", "wk", "wv")):
        close(r["g_" + k], leaves[1 + i].grad, what=k)
    if rel:
        close(r["g_pe_w"], leaves[4].grad, tol=1e-4, what="pe_w"); close(r["g_pe_b"], leaves[5].grad, tol=1e-4, what="pe_b")


@pytest.mark.parametrize("Bn,lens,E,R,layers", [(3, [5, 2, 7], 8, 8, 2), (1, [1], 4, 4, 1), (4, [1, 9, 4, 9], 16, 12, 2), (2, [6, 6], 32, 64, 3),
                                                (5, [3, 1, 2, 8, 5], 12, 20, 2)])
def test_language_f32_vs_torch_lstm_packed(Bn, lens, E, R, layers):
    """vog_lang_f32 against torch.nn.LSTM on packed sequences + the two Linear(+ReLU) layers and the start / end gather written
    with torch ops (NOT the oracle): forward activations and every gradient incl. the embedding rows."""
    g = torch.Generator().manual_seed(Bn * 77 + E + R)
    T, nsrl, V, D, L = max(lens), 3, 11, 6, 5
    sl = T + 2
    emb = torch.nn.Embedding(V + 1, E, padding_idx=V)
    lstm = torch.nn.LSTM(E, R, num_layers=layers, bidirectional=True, batch_first=True)
    proj, arg = torch.nn.Linear(2 * R, D), torch.nn.Linear(2 * D, L)
    with torch.no_grad():
        for p in list(emb.parameters()) + list(lstm.parameters()) + list(proj.parameters()) + list(arg.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    words = torch.randint(0, V, (Bn, 1, nsrl, sl), generator=g)
    mask = torch.full((Bn, 1, sl), -1, dtype=torch.int64)
    for b, ln in enumerate(lens):
        mask[b, 0, :ln] = torch.randint(0, nsrl * sl, (ln,), generator=g)
    cap = torch.stack([torch.stack([torch.sort(torch.randint(0, ln, (2,), generator=g)).values for _ in range(nsrl)]) for ln in lens]).unsqueeze(1)
    d_le = torch.randn(Bn * nsrl, L, generator=g)
    # torch reference
    wflat = words.reshape(Bn, nsrl * sl)
    m = mask.reshape(Bn, sl)
    tok = torch.where(m < 0, torch.full_like(m, V), torch.gather(wflat, 1, m.clamp(min=0)))[:, :T]
    x = emb(tok)
    pk = torch.nn.utils.rnn.pack_padded_sequence(x, torch.tensor(lens), batch_first=True, enforce_sorted=False)
    out, _ = lstm(pk)
    out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    full = torch.relu(proj(out))
    c2 = cap.reshape(Bn, nsrl, 2)
    st = torch.gather(full, 1, c2[..., 0].unsqueeze(-1).expand(-1, -1, D))
    en = torch.gather(full, 1, c2[..., 1].unsqueeze(-1).expand(-1, -1, D))
    le = torch.relu(arg(torch.cat([st, en], -1))).reshape(Bn * nsrl, L)
    (le * d_le).sum().backward()
    sd = {"lstm_encoder.embed_tokens.weight": emb.weight, "lstm_out_feat_proj.0.weight": proj.weight, "lstm_out_feat_proj.0.bias": proj.bias,
          "srl_arg_words_out_enc.0.weight": arg.weight, "srl_arg_words_out_enc.0.bias": arg.bias}
    for n_, p_ in lstm.named_parameters():
        sd["lstm_encoder.lstm." + n_] = p_
    batch = {"srl_arg_words_ind": words.cuda(), "srl_arg_word_mask": mask.cuda(), "srl_arg_word_mask_len": torch.tensor(lens).reshape(Bn, 1).cuda(),
             "srl_arg_words_capture": cap.cuda()}
    r = bwd.language_backward({k: v.detach() for k, v in sd.items()}, batch, T, layers, d_lang_enc=d_le.cuda())
    torch.cuda.synchronize()
    close(r["_lang_enc"], le, what="lang_enc"); close(r["_full"], full.reshape(Bn * T, D), what="full")
    for k, p_ in sd.items():
        close(r[k], p_.grad if p_.grad is not None else torch.zeros_like(p_), tol=1e-4, what=k)
