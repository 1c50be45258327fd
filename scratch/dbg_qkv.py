import sys; sys.path.insert(0, '.')
import ctypes as C, torch, math
from tests.gpu_util import L, t16
from oracle import vog_oracle as vo
lib = L.load()
torch.manual_seed(5)
S, N, H, d = 3, 37, 3, 32
heads = vo.chunk_sizes(d, H); dp, npad = 32, 64
td = torch.bfloat16
x = torch.randn(S * N, d, device="cuda").to(td)
wq, wk, wv = (torch.randn(d, d, device="cuda") / 6 for _ in range(3))
wpad = torch.zeros(3 * H * dp, d, device="cuda")
off = 0
for h, dh in enumerate(heads):
    for which, w in enumerate((wq, wk, wv)):
        wpad[(which * H + h) * dp:(which * H + h) * dp + dh] = w[off:off + dh]
    off += dh
wpad = wpad.to(td)
q = torch.full((S, H, N, dp), float("nan"), device="cuda").to(td)
k = torch.full((S, H, N, dp), float("nan"), device="cuda").to(td)
vt = torch.zeros((S, H, dp, npad), device="cuda").to(td)
print('ptrs', hex(q.data_ptr()), hex(k.data_ptr()), hex(vt.data_ptr()), q.numel()*2, vt.numel()*2)
a = L.QkvArgs()
a.x16, a.ldx, a.wqkv, a.ldw = L.ptr(x), d, L.ptr(wpad), d
a.q, a.k, a.vt = L.ptr(q), L.ptr(k), L.ptr(vt)
a.S, a.N, a.H, a.dp, a.npad, a.K, a.dtype = S, N, H, dp, npad, d, 0
L.check(lib.vog_qkv_proj(C.byref(a), L.stream_ptr()), "qkv")
torch.cuda.synchronize()
full = (x.float() @ wpad.float().t()).view(S, N, 3, H, dp)
kr = full[:, :, 1].permute(0, 2, 1, 3)
bad = (k.float() - kr).abs() > 2e-2
print('bad count', int(bad.sum()), 'of', bad.numel())
idx = bad.nonzero()
print(idx[:10].tolist(), idx[-5:].tolist())
# flat positions
flat = bad.reshape(-1).nonzero().view(-1)
print('flat range', int(flat.min()), int(flat.max()))
vr = full[:, :, 2].permute(0, 2, 3, 1)
print('vt err', (vt.float()[..., :N] - vr).abs().max().item(), 'vt pad nonzero', (vt[..., N:] != 0).sum().item())
print('q err', (q.float() - full[:, :, 0].permute(0,2,1,3)).abs().max().item())
# does bad region of k equal some vt rows?
kf = k.float().reshape(-1)
vtf = vr.reshape(S, H, dp, N)
print('k[0,0,0]', k[0,0,0].float().tolist()[:8])
print('vt row candidates', vr[0,0,0,:8].tolist())
