import torch
d=torch.device('cuda',0)
def t(fn,n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b)/n*1e3
for name,N,K,C in (('256->256 @32',2048,256,256),('512->512 @16',512,512,512),('128->128 @64',8192,128,128),('64->64 @128',32768,64,64)):
    V=torch.randn(16,N,K,device=d); U=torch.randn(16,K,C,device=d); M=torch.empty(16,N,C,device=d)
    tb=t(lambda: torch.bmm(V,U,out=M))
    V2=torch.randn(16,K,N,device=d); U2=torch.randn(16,C,K,device=d); M2=torch.empty(16,C,N,device=d)
    tb2=t(lambda: torch.bmm(U2,V2,out=M2))
    src=torch.empty(int((V.numel()+M.numel())*1.25)//1,device=d); dst=torch.empty_like(src)
    tc=t(lambda: dst.copy_(src))
    print(f'{name}: bmm NHWC-form {tb:.1f} us ({16*2*N*K*C/tb/1e6:.1f} TF), NCHW-form {tb2:.1f} us, transforms-as-copy {tc:.1f} us', flush=True)
