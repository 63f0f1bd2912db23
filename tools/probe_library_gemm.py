import torch, time
torch.backends.cuda.matmul.allow_bf16_reduced_precision_reduction=True
M=81216
def t(fn,n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for (N,K,nm) in [(1536,512,"QKV"),(512,512,"out-proj"),(2816,512,"FF1"),(512,1408,"FF2"),(512,1536,"dX qkv"),(1408,512,"dg"),(512,2816,"dX ff1")]:
    A=torch.randn(M,K,device="cuda",dtype=torch.bfloat16); B=torch.randn(N,K,device="cuda",dtype=torch.bfloat16)
    us=t(lambda: torch.matmul(A,B.t()))
    print(f"{nm:10s} M={M} N={N} K={K}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s (bf16 out, torch.matmul -> hipBLASLt)")
# weight gradient: dW = dY^T X  (N x K, reduce over M)
for (N,K,nm) in [(1536,512,"dW qkv"),(2816,512,"dW ff1"),(512,1408,"dW ff2")]:
    dY=torch.randn(M,N,device="cuda",dtype=torch.bfloat16); X=torch.randn(M,K,device="cuda",dtype=torch.bfloat16)
    us=t(lambda: torch.matmul(dY.t(),X))
    print(f"{nm:10s} N={N} K={K} R={M}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s")
