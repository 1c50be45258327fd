import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch, math, json
from tests.gpu_util import L
lib = L.load()
def run(M, N, K, residual=False, out32=True, out16=False, relu=False, bias=False, iters=200, dtype=torch.bfloat16):
    a = torch.randn(M, K, device='cuda').to(dtype); w = (torch.randn(N, K, device='cuda')/math.sqrt(K)).to(dtype)
    g = L.GemmArgs(); g.a, g.a_is_f32, g.lda = a.data_ptr(), 0, K; g.w, g.ldw = w.data_ptr(), K
    res = torch.randn(M, N, device='cuda') if residual else None
    b = torch.randn(N, device='cuda') if bias else None
    c32 = torch.empty(M, N, device='cuda') if out32 else None
    c16 = torch.empty(M, N, device='cuda', dtype=dtype) if out16 else None
    g.bias, g.residual, g.ldr = L.ptr(b), L.ptr(res), N
    g.c32, g.c16, g.ldc, g.ldc16 = L.ptr(c32), L.ptr(c16), N, N
    g.M, g.N, g.K, g.relu, g.rep, g.dtype = M, N, K, int(relu), 1, 0 if dtype == torch.bfloat16 else 1
    sp = L.stream_ptr()
    for _ in range(5): lib.vog_gemm_bias_act(C.byref(g), sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): lib.vog_gemm_bias_act(C.byref(g), sp)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / iters
    return us, 2.0 * M * N * K / us / 1e6
for name, kw in [("wo 4000x768x768 +res f32", dict(M=4000, N=768, K=768, residual=True)),
                 ("ffn1 4000x384x768 relu 16", dict(M=4000, N=384, K=768, out32=False, out16=True, relu=True, bias=True)),
                 ("ffn2 4000x768x384 +res", dict(M=4000, N=768, K=384, residual=True, bias=True)),
                 ("lin2 4000x256x768", dict(M=4000, N=256, K=768, relu=True, bias=True)),
                 ("qkvlike 4000x2304x768 16", dict(M=4000, N=2304, K=768, out32=False, out16=True)),
                 ("big 8192x8192x4096 16", dict(M=8192, N=8192, K=4096, out32=False, out16=True, iters=20)),
                 ("prop 800x256x2048", dict(M=800, N=256, K=2048, relu=True, bias=True)),
                 ("objqkv-like 800x1728x512 16", dict(M=800, N=1728, K=512, out32=False, out16=True)),
                 ]:
    us, tf = run(**kw)
    print(f"dbg={os.environ.get('VOG_GEMM_DEBUG','0')} {name:32s} {us:8.2f} us {tf:8.1f} TF")
