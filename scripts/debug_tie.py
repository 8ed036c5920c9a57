import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attention_based_tbn_amd._lib import call, ptr
DEV="cuda"
def st(): return torch.cuda.current_stream().cuda_stream
n,c,h,w = 3,128,2,8
row = torch.rand(n,c,1,w); row[row<0.4]=0
x = row.repeat(1,1,2,1).contiguous()
xr = x.double().requires_grad_(True)
yr = F.max_pool2d(xr,3,1,1,ceil_mode=True); dy = torch.randn(yr.shape); yr.backward(dy.double())
xd = x.permute(0,2,3,1).contiguous().to(DEV)
y = torch.empty(n,h,w,c,device=DEV); am = torch.empty(n*h*w*c,dtype=torch.uint8,device=DEV)
call("tbn_maxpool3_fwd", ptr(xd), c, ptr(y), c, ptr(am), n,h,w,c,h,w,1,1, st())
dx = torch.empty(n,h,w,c,device=DEV); dyd = dy.permute(0,2,3,1).contiguous().to(DEV)
call("tbn_maxpool3_bwd", ptr(dyd), c, ptr(am), ptr(dx), c, n,h,w,c,h,w,1,1,0, st())
e = (dx.permute(0,3,1,2).cpu().double()-xr.grad).abs().max()
print("tie test max err", float(e))
print(am.view(n,h,w,c)[0,:,:,0])
