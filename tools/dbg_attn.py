import torch, sys
sys.path.insert(0,'.')
from mvlt_amd import ops
torch.manual_seed(0)
def ref(q,kv,H,scale):
    B,N,C=q.shape; M=kv.shape[1]; hd=64
    qh=q.float().reshape(B,N,H,hd).permute(0,2,1,3)
    k=kv.float()[...,:C].reshape(B,M,H,hd).permute(0,2,1,3)
    v=kv.float()[...,C:].reshape(B,M,H,hd).permute(0,2,1,3)
    s=(qh@k.transpose(-1,-2))*scale
    return (s.softmax(-1)@v).transpose(1,2).reshape(B,N,C), torch.logsumexp(s,-1)
for dtype in (torch.bfloat16, torch.float32):
  for (B,H,N,M) in [(1,1,64,272),(1,1,64,256),(1,1,64,288),(1,1,64,224),(1,1,64,192),(1,1,64,320),(1,1,64,257)]:
    if dtype==torch.float32 and M>288: continue
    C=64*H
    q=torch.randn(B,N,C,device='cuda').to(dtype); kv=torch.randn(B,M,2*C,device='cuda').to(dtype)
    o=torch.empty_like(q); lse=torch.empty(B,H,N,device='cuda')
    ops.sr_attention_fwd(q,kv,o,lse,B,H,N,M,C,2*C,C,0,C,0.125)
    r,rl=ref(q,kv,H,0.125)
    err=(o.float()-r).abs()
    print(dtype,B,H,N,M,'maxerr',err.max().item(),'lse err',(lse-rl).abs().max().item(), 'bad d cols', (err.amax(dim=(0,1))>0.05).nonzero().flatten().tolist()[:20], 'bad q rows', (err.amax(dim=(0,2))>0.05).nonzero().flatten().tolist()[:10])
