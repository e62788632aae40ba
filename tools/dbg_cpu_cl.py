import sys, torch
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import jg_oracle as O
from joligen_amd.modules.resnet_generator import ResnetGenerator
net=ResnetGenerator(3,3,16,n_blocks=2)
sd={k:v.half().float() for k,v in O.synth_state_dict(net.state_dict(),0).items()}
g=torch.Generator().manual_seed(3)
x=(torch.rand(1,3,32,32,generator=g)*2-1).half().float()
Pm={k:v.clone().requires_grad_(True) for k,v in sd.items()}
fo=O.resnet_encoder(Pm,x,2,[10])[1][0]
keys=[k for k in Pm if k.endswith('weight') and k.startswith('encoder') and int(k.split('.')[2])<=10]
C,H,W=fo.shape[1:]
Rp=torch.randn(1,H*W,C,generator=g)
refs=torch.autograd.grad(fo,[Pm[k] for k in keys],Rp.view(1,H,W,C).permute(0,3,1,2).contiguous(),retain_graph=True)
(fo.permute(0,2,3,1).flatten(1,2)*Rp).sum().backward(retain_graph=True)
print(torch.__version__, torch.get_num_threads())
print(' '.join('%.4f'%float((Pm[k].grad-r).norm()/r.norm()) for k,r in zip(keys,refs)))
refs2=torch.autograd.grad(fo,[Pm[k] for k in keys],Rp.view(1,H,W,C).permute(0,3,1,2),retain_graph=True)
print(' '.join('%.4f'%float((a-r).norm()/r.norm()) for a,r in zip(refs2,refs)))
