"""p100 obj_tx attention alone (S=4, N=4000, H=3, dh=171 -> dp 192): time with / without the guard flag.
usage: python scratch/mb_attn_long.py [iters] [guard=1|0|both]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch, math
from tests.gpu_util import L
from tests.test_gpu_ops import to_frag
lib = L.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "both"
S, N, H, dh, dp = 4, 4000, 3, 171, 192
td = torch.bfloat16
npad = (N + 31)//32*32
torch.manual_seed(0)
q = torch.zeros(S,H,N,dp, device='cuda'); k = torch.zeros_like(q); v = torch.zeros_like(q)
q[..., :dh] = torch.randn(S,H,N,dh, device='cuda'); k[..., :dh] = torch.randn(S,H,N,dh, device='cuda'); v[..., :dh] = torch.randn(S,H,N,dh, device='cuda')
qf, kf, vf = to_frag(q.to(td),'qk'), to_frag(k.to(td),'qk'), to_frag(v.to(td),'v')
u = torch.randn(S, N, H, device='cuda'); peb = torch.randn(H, device='cuda')
out = torch.zeros(S*N, H*dp, device='cuda', dtype=td)
flag = torch.zeros(4, dtype=torch.int32, device='cuda')
def run(guard):
    a = L.AttnArgs()
    a.q, a.k, a.vt, a.out16, a.u, a.pe_b = L.ptr(qf), L.ptr(kf), L.ptr(vf), L.ptr(out), L.ptr(u), L.ptr(peb)
    a.S, a.N, a.H, a.dp, a.npad, a.use_rel, a.n_box, a.seq_per_vid, a.NP = S, N, H, dp, npad, 1, N, 1, N
    a.inv_scale, a.dtype = 1.0/math.sqrt(H*dh), 0
    a.guard_flag = L.ptr(flag) if guard else None
    sp = L.stream_ptr()
    for _ in range(3): lib.vog_rel_attention_fwd(C.byref(a), sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): lib.vog_rel_attention_fwd(C.byref(a), sp)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
fl = 4.0 * S * H * N * N * dh
for g in ((1, 0) if which == "both" else (int(which),)):
    t = run(g)
    print(f"guard={g}: {t:.1f} us  {fl / t / 1e6:.0f} TFLOP/s (head dim {dh})  flag={int(flag[0])}")
