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
wpad = (torch.randn(3 * H * dp, d, device="cuda") / 6).to(td)
SZ = 1 << 16
buf = torch.zeros(8 * SZ, dtype=td, device="cuda")
a = L.QkvArgs()
a.x16, a.ldx, a.wqkv, a.ldw = L.ptr(x), d, L.ptr(wpad), d
base = buf.data_ptr()
a.q, a.k, a.vt = base + 1 * SZ * 2, base + 3 * SZ * 2, base + 5 * SZ * 2
a.S, a.N, a.H, a.dp, a.npad, a.K, a.dtype = S, N, H, dp, npad, d, 0
L.check(lib.vog_qkv_proj(C.byref(a), L.stream_ptr()), "qkv")
torch.cuda.synchronize()
nz = (buf != 0).nonzero().view(-1)
segs = {}
for i in nz.tolist():
    segs.setdefault(i // SZ, [0, 1 << 30, 0])
    s = segs[i // SZ]; s[0] += 1; s[1] = min(s[1], i % SZ); s[2] = max(s[2], i % SZ)
print('segments (idx: count, min, max):', segs)
full = (x.float() @ wpad.float().t()).view(S, N, 3, H, dp)
for name, seg, ref in (('q', 1, full[:, :, 0].permute(0, 2, 1, 3)), ('k', 3, full[:, :, 1].permute(0, 2, 1, 3))):
    got = buf[seg * SZ: seg * SZ + S * H * N * dp].float().view(S, H, N, dp)
    print(name, 'err', (got - ref).abs().max().item())
got = buf[5 * SZ: 5 * SZ + S * H * dp * npad].float().view(S, H, dp, npad)[..., :N]
print('vt err', (got - full[:, :, 2].permute(0, 2, 3, 1)).abs().max().item())
