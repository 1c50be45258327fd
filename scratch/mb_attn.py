import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch, math
from tests.gpu_util import L
from tests.test_gpu_ops import to_frag
lib = L.load()
def run(S, N, H, dh, dp, nsrl, iters=100):
    td = torch.bfloat16
    npad = (N + 31)//32*32
    q = torch.zeros(S,H,N,dp, device='cuda'); k = torch.zeros_like(q); v = torch.zeros_like(q)
    q[..., :dh] = torch.randn(S,H,N,dh, device='cuda'); k[..., :dh] = torch.randn(S,H,N,dh, device='cuda'); v[..., :dh] = torch.randn(S,H,N,dh, device='cuda')
    qf, kf, vf = to_frag(q.to(td),'qk'), to_frag(k.to(td),'qk'), to_frag(v.to(td),'v')
    n_box = N//nsrl
    u = torch.randn(S, n_box, H, device='cuda'); peb = torch.randn(H, device='cuda')
    out = torch.zeros(S*N, H*dp, device='cuda', dtype=td)
    a = L.AttnArgs()
    a.q, a.k, a.vt, a.out16, a.u, a.pe_b = L.ptr(qf), L.ptr(kf), L.ptr(vf), L.ptr(out), L.ptr(u), L.ptr(peb)
    a.S, a.N, a.H, a.dp, a.npad, a.use_rel, a.n_box, a.seq_per_vid, a.NP = S, N, H, dp, npad, 1, n_box, 1, n_box
    a.inv_scale, a.dtype = 1.0/math.sqrt(H*dh), 0
    sp = L.stream_ptr()
    for _ in range(5): lib.vog_rel_attention_fwd(C.byref(a), sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): lib.vog_rel_attention_fwd(C.byref(a), sp)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
print('mul 40x100x3x256', run(40, 100, 3, 256, 256, 5))
print('obj 4x200x3x171 ', run(4, 200, 3, 171, 192, 1))
