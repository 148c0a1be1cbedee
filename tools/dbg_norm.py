import torch, sys
sys.path.insert(0, '.')
from pixelwiseregression_amd import kernels as K
dev='cuda:0'
for dt in (torch.float32, torch.bfloat16):
    for (B,H,W,C) in ((2,64,64,128),(3,2,2,64),(2,5,7,16)):
        y=torch.randn(B,H,W,C,device=dev).to(dt); g=torch.randn(B,H,W,C,device=dev).to(dt)
        st=K.norm_stats(y, torch.ones(C,device=dev), torch.zeros(C,device=dev))
        torch.cuda.synchronize(); print('stats ok', dt, B,H,W,C, flush=True)
        a=K.norm_bwd(g,y,st); torch.cuda.synchronize(); print('bwd ok', flush=True)
        b=K.norm_bwd_split(g,y,st); torch.cuda.synchronize(); print('split ok', (a[0].float()-b[0].float()).abs().max().item(), (a[1]-b[1]).abs().max().item(), flush=True)
