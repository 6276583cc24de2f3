import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import mhla_amd
from mhla_amd import _lib
from oracle import mhla_oracle as orc
from gpu_util import make_blockmix_inputs
B,H,M,S,D = 1,1,8,64,64
q,k,v,W,do,_,_ = make_blockmix_inputs(B,H,M,S,D, torch.bfloat16, w="rand")
out, aux = orc.blockmix_fwd(q.float(),k.float(),v.float(),W, return_aux=True)
lib = _lib.load()
from mhla_amd.ops import _view, _ws
qd,kd,vd,Wd = (t.cuda() for t in (q,k,v,W))
o = torch.empty_like(qd)
nbytes = lib.mhla_blockmix_fwd_ws_bytes(B,H,M,S,D,1,0,0)
ws = _ws(nbytes, qd.device); ws.zero_()
rc = lib.mhla_blockmix_fwd(_view(qd),_view(kd),_view(vd),_view(qd),_view(kd),Wd.data_ptr(),M,_view(o),None,ws.data_ptr(),ws.numel()*4,B,H,M,S,D,1,1e-6,0,torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); print("rc", rc)
njg = (M+7)//8
st_floats = B*H*njg*4096*8*2//4
z = ws[st_floats:st_floats+B*H*M*S].cpu().reshape(M,S)
ks = ws[st_floats+B*H*M*S: st_floats+B*H*M*S+B*H*M*64].cpu().reshape(M,64)
print("z err", (z-aux["z"][0]).abs().max().item(), aux["z"].abs().max().item())
print("ksum err", (ks-aux["ksum"][0]).abs().max().item(), aux["ksum"].abs().max().item())
print(z[0,:8], aux["z"][0,0,:8]); print(ks[0,:8], aux["ksum"][0,0,:8])
off = st_floats+B*H*M*S+B*H*M*64
ninv = ws[off:off+B*H*M*S].cpu().reshape(M,S)
print("ninv err", (ninv - 1.0/aux["n"][0]).abs().max().item(), (1.0/aux["n"]).abs().max().item())
print("out err", (o.float().cpu()-out).abs().max().item(), out.abs().max().item())
st = ws[:st_floats].view(torch.bfloat16).float().cpu().reshape(njg,64,64,8)   # [jg][d2][d1][jj]
kvT = aux["kv"][0].transpose(-1,-2)   # [M][d2][d1]
print("state err", (st[0].permute(2,0,1) - kvT[:8]).abs().max().item(), kvT.abs().max().item())
