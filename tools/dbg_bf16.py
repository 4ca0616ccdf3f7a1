import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import filler, pvlt_oracle as O
from mvlt_amd import pvlt
name='tiny256_pretrain'
g=np.load(f'tests/golden/{name}.npz')
seed,B,img,T,dp=int(g['meta'][0]),int(g['meta'][1]),int(g['meta'][2]),int(g['meta'][3]),float(g['meta'][4])
lt=dict(mlm=1,itm=1,t2i=1,cls=0)
cfg=O.Cfg('pvlt_tiny',lt,224,768,T,dp)
sd=O.filled_state_dict(cfg,seed)
batch=O.to_torch_batch(filler.make_batch(seed,B,img,T))
def sample(t,n=64):
    f=t.detach().reshape(-1).float().cpu(); st=max(1,f.numel()//n); return f[::st][:n].numpy()
dev=torch.device('cuda:0')
for dtype in (torch.float32, torch.bfloat16):
    m=pvlt.pvlt_tiny(pretrained=False,token_hidden_size=768,num_text_tokens=T,loss_type=lt,pretrained_pth=None,drop_path_rate=dp,compute_dtype=dtype)
    m.load_state_dict(sd); m.cuda().eval(); m._taps={}
    with torch.no_grad(): out=m(batch['image'].to(dev),batch['input_ids'].to(dev))
    print(dtype)
    for k in sorted(g.files):
        if k.startswith('eval/tap/') and k.endswith('/sample'):
            tap=k.split('/')[2]
            if tap in m._taps:
                a=sample(m._taps[tap]); b=g[k]
                print(f'  {tap:12s} maxnorm {np.abs(a-b).max()/np.abs(b).max():.4f}  relL2 {np.linalg.norm(a-b)/np.linalg.norm(b):.4f}  ref absmax {np.abs(b).max():.3f} rms {np.sqrt((b*b).mean()):.3f}')
    for key in ('mlm_logits','itm_logits','t2i_logits'):
        a=sample(out[key].float(),256); b=g[f'eval/out/{key}/sample']
        print(f'  {key:12s} maxnorm {np.abs(a-b).max()/np.abs(b).max():.4f}  relL2 {np.linalg.norm(a-b)/np.linalg.norm(b):.4f}')
# ---- full-tensor look at stage 4 text tokens (oracle live)
from oracle.hostinfo import usable_cores
torch.set_num_threads(usable_cores())
taps={}
with torch.no_grad(): ref=O.forward(sd,cfg,batch['image'],batch['input_ids'],taps=taps)
a=m._taps['text_feat4'].cpu(); b=taps['text_feat4']
err=((a-b).norm(dim=-1)/b.norm(dim=-1))
print('text_feat4 per-token relerr: mean',err.mean().item(),'max',err.max().item())
ids=batch['input_ids']
for bb in range(2):
    print(' b',bb,'ids[:12]',ids[bb,:12].tolist(),'... err[:12]',[round(x,3) for x in err[bb,:12].tolist()], 'err[60:70]',[round(x,3) for x in err[bb,60:70].tolist()], 'err[-5:]',[round(x,3) for x in err[bb,-5:].tolist()])
print(' token norms ref[:8]', [round(x,3) for x in b[0,:8].norm(dim=-1).tolist()], ' pad norms', [round(x,3) for x in b[0,-4:].norm(dim=-1).tolist()])
a3=m._taps['text_feat3'].cpu(); b3=taps['text_feat3']
e3=((a3-b3).norm(dim=-1)/b3.norm(dim=-1)); print('text_feat3 per-token relerr mean',e3.mean().item(),'max',e3.max().item())
ai=m._taps['img_feat4'].cpu(); bi=taps['img_feat4']
ei=((ai-bi).flatten(2).norm(dim=1)/bi.flatten(2).norm(dim=1)); print('img_feat4 per-token relerr mean',ei.mean().item(),'max',ei.max().item())
print('overall relL2 text4', ((a-b).norm()/b.norm()).item(), 'img4', ((ai-bi).norm()/bi.norm()).item())
