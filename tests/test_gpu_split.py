"""-m gpu: the hi + lo operand kernels of round 6 (three MFMAs per product: x.w + x_lo.w + x.w_lo with x_lo = t16(x - t16(x)))
against fp32 torch references on the UNROUNDED operands: what they exist for is that the result does not carry the 2^-11 operand
rounding of the plain 16-bit kernels. Tolerances are written per test; every case also runs the plain kernel on the same data to
show the difference the second operand makes."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from tests.gpu_util import L, t16
from tests.test_gpu_ops import DT, _attn_ref, _lib, _ln, _pack32, _sp, from_frag, frag_index, to_frag

pytestmark = pytest.mark.gpu


def hi_lo(x, dtype="f16"):
    td = t16(dtype)
    hi = x.to(td)
    return hi, (x - hi.float()).to(td)


@pytest.mark.parametrize("S,N,H,dh,dp,nsrl,use_rel", [
    (4, 200, 3, 171, 192, 1, 1), (40, 100, 3, 256, 256, 5, 1), (6, 25, 3, 256, 256, 5, 1), (16, 50, 3, 171, 192, 1, 1),
    (3, 140, 2, 64, 64, 1, 0), (2, 256, 1, 128, 128, 1, 1), (2, 33, 2, 11, 32, 1, 1)])
def test_rel_attention_hi_lo(S, N, H, dh, dp, nsrl, use_rel):
    """Sharp attention (logit std ~ 20 nats): Q.K^T from hi + lo fragments follows the fp32 logits; the plain f16 kernel on the
    same data is off by whole percent on competing probabilities. Also: out16 + out16_lo carries the output to ~2^-20 and
    logit_max reports the largest |logit| of the launch."""
    lib = _lib()
    torch.manual_seed(S * 1000 + N)
    npad = (N + 31) // 32 * 32
    q = torch.zeros(S, H, N, dp, device="cuda"); k = torch.zeros_like(q); v = torch.zeros_like(q)
    sc = math.sqrt(20.0 * math.sqrt(H * dh) / math.sqrt(dh))          # logits / sqrt(H dh) with std ~ 20
    q[..., :dh] = torch.randn(S, H, N, dh, device="cuda") * sc
    k[..., :dh] = torch.randn(S, H, N, dh, device="cuda") * sc
    v[..., :dh] = torch.randn(S, H, N, dh, device="cuda")
    (qh, ql), (kh, kl) = hi_lo(q), hi_lo(k)
    v16 = v.to(torch.float16)
    n_box = N // nsrl
    u_box = torch.randn(S, n_box, H, device="cuda") * 3
    peb = torch.randn(H, device="cuda")
    inv_scale = 1.0 / math.sqrt(H * dh)
    u_tok = u_box.repeat(1, nsrl, 1)
    ref = _attn_ref(q, k, v16.float(), u_tok, peb, n_box, inv_scale, use_rel)      # fp32 q, k
    outs = {}
    for split in (0, 1):
        out = torch.full((S * N, H * dp), float("nan"), device="cuda").to(torch.float16)
        out_lo = torch.full_like(out, float("nan"))
        lmax = torch.zeros(L.LOGIT_WORDS * L.LOGIT_STRIDE, dtype=torch.int32, device="cuda")   # (vog_attn_args.logit_max: 4 KiB)
        a = L.AttnArgs()
        frs = [to_frag(qh, "qk"), to_frag(kh, "qk"), to_frag(v16, "v")]       # (kept alive: the kernel reads them)
        a.q, a.k, a.vt, a.out16 = L.ptr(frs[0]), L.ptr(frs[1]), L.ptr(frs[2]), L.ptr(out)
        a.u, a.pe_b = L.ptr(u_box.contiguous()), L.ptr(peb)
        a.S, a.N, a.H, a.dp, a.npad = S, N, H, dp, npad
        a.use_rel, a.n_box, a.seq_per_vid, a.NP = use_rel, n_box, 1, n_box
        a.inv_scale, a.dtype = inv_scale, DT["f16"]
        a.logit_max = L.ptr(lmax)
        keep = []
        if split:
            keep = [to_frag(ql, "qk"), to_frag(kl, "qk")]
            a.q_lo, a.k_lo, a.out16_lo = L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(out_lo)
        L.check(lib.vog_rel_attention_fwd(C.byref(a), _sp()), "attn")
        torch.cuda.synchronize()
        got = out.float().view(S, N, H, dp).permute(0, 2, 1, 3)
        assert torch.isfinite(got).all()
        outs[split] = (got - ref).abs().max().item()
        if split:
            full = (out.float() + out_lo.float()).view(S, N, H, dp).permute(0, 2, 1, 3)
            e2 = (full - ref).abs().max().item()
            # P and V are still f16 (2^-11 each, averaged by the sum): 2e-3 of the value scale
            assert e2 <= 2e-3 * max(1.0, ref.abs().max().item()), e2
            assert (full[..., dh:] == 0).all()
        lg = (q @ k.transpose(-1, -2))
        if use_rel:
            ub = u_tok.permute(0, 2, 1)
            lg = lg + torch.relu(ub.unsqueeze(-1) - ub.unsqueeze(-2) + peb.view(1, -1, 1, 1))
        want = (lg * inv_scale).abs().max().item()
        seen = float(lmax.cpu().numpy().view(np.float32).max())
        assert abs(seen - want) <= (2e-2 if not split else 1e-3) * want, (seen, want)
    print(f"max abs error vs fp32 logits: plain f16 {outs[0]:.2e}, hi + lo {outs[1]:.2e}")
    assert outs[1] <= 2.5e-3 * max(1.0, ref.abs().max().item())
    assert outs[0] > 3 * outs[1]            # the case is sharp enough to tell the two apart


@pytest.mark.parametrize("S,nfrm,nsrl,nppf,H,dh,dp,use_rel,lpv", [
    (40, 10, 5, 20, 3, 256, 256, 1, 0), (6, 3, 5, 5, 3, 256, 256, 1, 1), (8, 4, 5, 20, 3, 128, 128, 1, 0), (4, 4, 3, 32, 1, 64, 64, 0, 0)])
def test_rel_attention_struct_hi_lo(S, nfrm, nsrl, nppf, H, dh, dp, use_rel, lpv):
    """Separable mul_tx layer-0 attention with hi + lo visual Q / K parts (language parts split in the kernel from fp32) at a
    logit std of ~ 20 nats: equals the full fp32 softmax over all (a', p') keys."""
    lib = _lib()
    torch.manual_seed(S * 100 + nppf)
    n_vid = S // nfrm
    n_lang = n_vid if lpv else 1
    nc_v = 1 if lpv else n_vid
    Nq, hd = nsrl * nppf, H * dp
    npad_kv = (nppf + 31) // 32 * 32
    sc = math.sqrt(20.0 * math.sqrt(H * dh) / math.sqrt(dh) / 2.0)
    qv = torch.zeros(S, H, nppf, dp, device="cuda"); kvv = torch.zeros_like(qv); vvv = torch.zeros_like(qv)
    qv[..., :dh] = torch.randn(S, H, nppf, dh, device="cuda") * sc
    kvv[..., :dh] = torch.randn(S, H, nppf, dh, device="cuda") * sc
    vvv[..., :dh] = torch.randn(S, H, nppf, dh, device="cuda")
    pl = torch.zeros(n_lang * nsrl, 3, H, dp, device="cuda")
    pl[..., :dh] = torch.randn(n_lang * nsrl, 3, H, dh, device="cuda")
    pl[:, :2] *= sc
    lrow = torch.tensor([(s // nfrm) if lpv else (s // nfrm) // nc_v for s in range(S)], device="cuda")
    pls = pl.view(n_lang, nsrl, 3, H, dp)[lrow]
    ql, kl, vl = (pls[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    (qh, qlo), (kh, klo) = hi_lo(qv), hi_lo(kvv)
    vv16 = vvv.to(torch.float16)
    u_box = torch.randn(n_vid, nfrm * nppf, H, device="cuda") * 2
    peb = torch.randn(H, device="cuda")
    inv_scale = 1.0 / math.sqrt(H * dh)
    plc = pl.reshape(n_lang * nsrl, 3 * hd).contiguous()
    # full fp32 reference (values as the kernel sees them: f16)
    q_tok = (qv.unsqueeze(2) + ql.unsqueeze(3)).reshape(S, H, Nq, dp)
    k_tok = (kvv.unsqueeze(2) + kl.unsqueeze(3)).reshape(S, H, Nq, dp)
    v_tok = (vv16.float().unsqueeze(2) + vl.to(torch.float16).float().unsqueeze(3)).reshape(S, H, Nq, dp)
    logits = q_tok @ k_tok.transpose(-1, -2)
    if use_rel:
        ub = u_box.view(n_vid, nfrm, nppf, H)[torch.arange(S, device="cuda") // nfrm, torch.arange(S, device="cuda") % nfrm]
        ut = ub.repeat(1, nsrl, 1).permute(0, 2, 1)
        logits = logits + torch.relu(ut.unsqueeze(-1) - ut.unsqueeze(-2) + peb.view(1, -1, 1, 1))
    ref = torch.softmax(logits * inv_scale, dim=-1) @ v_tok
    errs = {}
    for split in (0, 1):
        out = torch.full((S * Nq, hd), float("nan"), device="cuda").to(torch.float16)
        lmax = torch.zeros(L.LOGIT_WORDS * L.LOGIT_STRIDE, dtype=torch.int32, device="cuda")   # (vog_attn_args.logit_max: 4 KiB)
        a = L.AttnStructArgs()
        keep = [to_frag(qh, "qk"), to_frag(kh, "qk"), to_frag(vv16, "v"), to_frag(qlo, "qk"), to_frag(klo, "qk")]
        a.q_visual = 1
        a.q, a.kv, a.vv, a.pl, a.out16 = L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(keep[2]), L.ptr(plc), L.ptr(out)
        a.u, a.pe_b = L.ptr(u_box), L.ptr(peb)
        a.S, a.H, a.dp, a.nsrl, a.nppf, a.npad_q, a.npad_kv = S, H, dp, nsrl, nppf, (Nq + 31) // 32 * 32, npad_kv
        a.nfrm, a.lang_per_vid, a.nc_v = nfrm, lpv, nc_v
        a.use_rel, a.seq_per_vid, a.NP, a.inv_scale, a.dtype = use_rel, nfrm, nfrm * nppf, inv_scale, DT["f16"]
        a.logit_max = L.ptr(lmax)
        if split:
            a.q_lo, a.kv_lo = L.ptr(keep[3]), L.ptr(keep[4])
        L.check(lib.vog_rel_attention_struct_fwd(C.byref(a), _sp()), "struct attention")
        torch.cuda.synchronize()
        got = out.float().view(S, Nq, H, dp).permute(0, 2, 1, 3)
        assert torch.isfinite(got).all()
        errs[split] = (got - ref).abs().max().item()
        seen = float(lmax.cpu().numpy().view(np.float32).max())
        want = (logits * inv_scale).abs().max().item()
        assert want * 0.98 <= seen <= 2.05 * want, (seen, want)     # (a bound: max|x| + max|y| of the separable parts)
    print(f"struct attention, max abs error vs fp32: plain f16 {errs[0]:.2e}, hi + lo {errs[1]:.2e}")
    assert errs[1] <= 3e-3 * max(1.0, ref.abs().max().item())
    assert errs[0] > 3 * errs[1]


@pytest.mark.parametrize("S,N,H,d", [(4, 200, 3, 512), (40, 20, 3, 512), (5, 100, 3, 768), (3, 37, 3, 64)])
def test_qkv_proj_hi_lo(S, N, H, d):
    """x.w + x_lo.w + x.w_lo: Q / K come back as hi + lo fragments that add up to the fp32 projection (2^-20 of the row scale);
    V^T as one f16 image."""
    lib = _lib()
    torch.manual_seed(S + N)
    dh = (d + H - 1) // H
    dp = {22: 32, 171: 192, 256: 256}[dh]
    npad = (N + 31) // 32 * 32
    x = torch.randn(S * N, d, device="cuda")
    wpad = torch.zeros(3 * H * dp, d, device="cuda")
    for wh in range(3 * H):
        wpad[wh * dp: wh * dp + dh] = torch.randn(dh, d, device="cuda") / math.sqrt(d) * 4
    (xh, xl), (wh_, wl) = hi_lo(x), hi_lo(wpad)
    bufs = [torch.zeros((S, H, npad * dp), device="cuda").to(torch.float16) for _ in range(5)]
    a = L.QkvArgs()
    a.x16, a.ldx, a.wqkv, a.ldw = L.ptr(xh), d, L.ptr(wh_), d
    a.q, a.k, a.vt, a.q_lo, a.k_lo = (L.ptr(b) for b in bufs)
    a.x16_lo, a.wqkv_lo = L.ptr(xl), L.ptr(wl)
    a.S, a.N, a.H, a.dp, a.npad, a.K, a.dtype = S, N, H, dp, npad, d, DT["f16"]
    L.check(lib.vog_qkv_proj(C.byref(a), _sp()), "qkv hi + lo")
    torch.cuda.synchronize()
    full = (x.double() @ wpad.double().t()).float().view(S, N, 3, H, dp)
    scale = full.abs().max().item()
    for which, kind in ((0, "qk"), (1, "qk")):
        got = from_frag(bufs[which].float(), N, dp, kind) + from_frag(bufs[3 + which].float(), N, dp, kind)
        ref = full[:, :, which].permute(0, 2, 1, 3)
        err = (got - ref).abs().max().item()
        plain = (from_frag(bufs[which].float(), N, dp, kind) - ref).abs().max().item()
        assert err <= 4e-6 * scale, (which, err, scale)
        assert plain > 20 * err                                   # (the hi image alone is an f16 rounding away)
    gv = from_frag(bufs[2].float(), N, dp, "v")
    assert (gv - full[:, :, 2].permute(0, 2, 1, 3)).abs().max().item() <= 1e-3 * scale
    for b, kind in ((bufs[3], "qk"), (bufs[4], "qk")):            # nothing outside the valid slots was touched
        msk = torch.ones(npad * dp, dtype=torch.bool, device="cuda")
        msk[frag_index(N, dp, kind).reshape(-1).cuda()] = False
        assert (b[:, :, msk] == 0).all()


def _pack16(w, dtype="f16"):
    w = np.ascontiguousarray(w.detach().cpu().numpy(), dtype=np.float32)
    N, K = w.shape
    dst = np.empty(N * K, dtype=np.uint16)
    L.check(_lib().vog_pack_w_frag(w.ctypes.data, K, N, K, dst.ctypes.data, DT[dtype]), "pack")
    return torch.from_numpy(dst.view(np.int16)).cuda()


def test_gemm_skinny_hi_lo():
    """The language half of mul_tx's layer-0 QKV (M = 20, K = 256): fp32 rows split in the kernel, W + W_lo in fragment order."""
    lib = _lib()
    torch.manual_seed(3)
    M, N, K = 20, 2304, 256
    a32 = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / 8
    wh = w.to(torch.float16).float()
    wf, wlf = _pack16(wh), _pack16(w - wh)
    out = torch.zeros(M, N, device="cuda")
    g = L.GemmArgs()
    g.a, g.a_is_f32, g.lda, g.w, g.ldw, g.w_frag, g.w_lo = L.ptr(a32), 1, K, L.ptr(wf), K, 1, L.ptr(wlf)
    g.c32, g.ldc, g.M, g.N, g.K, g.rep, g.dtype = L.ptr(out), N, M, N, K, 1, DT["f16"]
    L.check(lib.vog_gemm_bias_act(C.byref(g), _sp()), "skinny hi + lo")
    torch.cuda.synchronize()
    ref = (a32.double() @ w.double().t()).float()
    err = (out - ref).abs().max().item()
    plain = (a32.to(torch.float16).float() @ wh.t() - ref).abs().max().item()
    assert err <= 4e-6 * ref.abs().max().item(), err
    assert plain > 20 * err


@pytest.mark.parametrize("rows,nppf0", [(800, 5), (75, 5), (1600, 10)])
def test_vis_encode_hi_lo(rows, nppf0):
    """Both feature encoders with hi + lo operands: c32 = relu(W x + b) to fp32 accuracy (the plain f16 form carries 2^-11
    per operand), c16 + c16_lo = c32."""
    lib = _lib()
    torch.manual_seed(rows)
    Kp, Ks, Np, Ns = 2048, 3072, 256, 256
    prop = torch.randn(rows, Kp, device="cuda")
    seg = torch.randn(rows // nppf0, Ks, device="cuda")
    wp = torch.randn(Np, Kp, device="cuda") / math.sqrt(Kp)
    wsg = torch.randn(Ns, Ks, device="cuda") / math.sqrt(Ks)
    bp, bs = torch.randn(Np, device="cuda") * 0.1, torch.randn(Ns, device="cuda") * 0.1
    wph, wsh = wp.to(torch.float16).float(), wsg.to(torch.float16).float()
    keep = [_pack16(wph), _pack16(wsh), _pack16(wp - wph), _pack16(wsg - wsh)]
    ref = torch.cat([torch.relu(prop.double() @ wp.double().t() + bp.double()),
                     torch.relu(seg.double() @ wsg.double().t() + bs.double()).repeat_interleave(nppf0, 0)], 1).float()
    errs = {}
    for split in (0, 1):
        c32 = torch.full((rows, Np + Ns), float("nan"), device="cuda")
        c16 = torch.zeros(rows, Np + Ns, device="cuda").to(torch.float16)
        c16l = torch.full_like(c16, float("nan"))
        a = L.VisencArgs()
        a.prop, a.seg, a.w_prop_f, a.w_seg_f, a.b_prop, a.b_seg = L.ptr(prop), L.ptr(seg), L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(bp), L.ptr(bs)
        a.c32, a.c16, a.ldc, a.c16_dtype = L.ptr(c32), L.ptr(c16), Np + Ns, DT["f16"]
        a.n_prop_rows, a.nppf0, a.prop_dim, a.seg_dim, a.prop_enc, a.seg_enc, a.dtype, a.lean = rows, nppf0, Kp, Ks, Np, Ns, DT["f16"], 1
        if split:
            a.w_prop_f_lo, a.w_seg_f_lo, a.c16_lo = L.ptr(keep[2]), L.ptr(keep[3]), L.ptr(c16l)
        L.check(lib.vog_vis_encode(C.byref(a), _sp()), "vis_encode")
        torch.cuda.synchronize()
        errs[split] = (c32 - ref).abs().max().item()
        if split:
            assert (c16.float() + c16l.float() - c32).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    print(f"encoders, max abs error vs fp64: plain f16 {errs[0]:.2e}, hi + lo {errs[1]:.2e}")
    assert errs[1] <= 5e-6 * max(1.0, ref.abs().max().item())
    assert errs[0] > 20 * errs[1]


@pytest.mark.parametrize("M,d,kwo", [(800, 512, 576), (1000, 768, 768), (70, 512, 576)])
def test_tx_tail_hi_lo(M, d, kwo):
    """Wo + residual + LN + FFN + residual + LN with hi + lo operands in all three GEMM stages: y32 equals the fp32 chain on the
    fp32 operands (the plain f16 kernel: 2^-11 per operand and stage); y16 + y16_lo = y32."""
    lib = _lib()
    torch.manual_seed(M + d)
    dh = d // 2
    attn = torch.randn(M, kwo, device="cuda")
    res = torch.randn(M, d, device="cuda")
    wo = torch.randn(d, kwo, device="cuda") / math.sqrt(kwo)
    w1 = torch.randn(dh, d, device="cuda") / math.sqrt(d)
    w2 = torch.randn(d, dh, device="cuda") / math.sqrt(dh)
    b1, b2 = torch.randn(dh, device="cuda") * 0.1, torch.randn(d, device="cuda") * 0.1
    g1, be1 = 1 + 0.1 * torch.randn(d, device="cuda"), 0.1 * torch.randn(d, device="cuda")
    g2, be2 = 1 + 0.1 * torch.randn(d, device="cuda"), 0.1 * torch.randn(d, device="cuda")
    x1 = _ln(res.double() + attn.double() @ wo.double().t(), g1.double(), be1.double())
    ref = _ln(x1 + torch.relu(x1 @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double(), g2.double(), be2.double()).float()
    (ah, al) = hi_lo(attn)
    hl = lambda w: (w.to(torch.float16).float(), w - w.to(torch.float16).float())
    (woh, wol), (w1h, w1l), (w2h, w2l) = hl(wo), hl(w1), hl(w2)
    keep = [_pack32(woh, "f16"), _pack32(w1h, "f16"), _pack32(w2h, "f16"), _pack32(wol, "f16"), _pack32(w1l, "f16"), _pack32(w2l, "f16")]
    errs = {}
    for split in (0, 1):
        y32 = torch.full((M, d), float("nan"), device="cuda")
        y16 = torch.zeros(M, d, device="cuda").to(torch.float16)
        y16l = torch.full_like(y16, float("nan"))
        a = L.TxTailArgs()
        a.attn16, a.kwo, a.wo_p, a.w1_p, a.w2_p = L.ptr(ah), kwo, L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(keep[2])
        a.residual, a.ldr = L.ptr(res), d
        a.ln1g, a.ln1b, a.b1, a.b2, a.ln2g, a.ln2b = L.ptr(g1), L.ptr(be1), L.ptr(b1), L.ptr(b2), L.ptr(g2), L.ptr(be2)
        a.y32, a.y16, a.M, a.d, a.dh, a.dtype = L.ptr(y32), L.ptr(y16), M, d, dh, DT["f16"]
        if split:
            a.attn16_lo, a.wo_p_lo, a.w1_p_lo, a.w2_p_lo, a.y16_lo = L.ptr(al), L.ptr(keep[3]), L.ptr(keep[4]), L.ptr(keep[5]), L.ptr(y16l)
        L.check(lib.vog_tx_tail_fwd(C.byref(a), _sp()), "tail")
        torch.cuda.synchronize()
        assert torch.isfinite(y32).all()
        errs[split] = (y32 - ref).abs().max().item()
        if split:
            assert (y16.float() + y16l.float() - y32).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    print(f"tail d = {d}, max abs error vs fp64: plain f16 {errs[0]:.2e}, hi + lo {errs[1]:.2e}")
    assert errs[1] <= 2e-5 * max(1.0, ref.abs().max().item())
    assert errs[0] > 20 * errs[1]
