import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import soap as osoap
from metatrain_amd.soap_bpnn import SoapBpnnHip
dev=torch.device("cuda:0")
types=[1,6,7,8]
for legacy in (True, False):
    hy=dict(osoap.DEFAULT_HYPERS, legacy=legacy)
    params=osoap.synthetic_params(hy,4,osoap.basis(hy)[0],0,torch.float32)
    m=SoapBpnnHip(hy,types); m.load({k:v.to(dev) for k,v in params.items()})
    for name,(pos,z,sysidx,ns) in {"empty":(np.zeros((0,3)),[],[],1), "isolated":(np.array([[0.,0,0],[50,0,0],[0,50,0]]),[1,6,8],[0,0,1],2)}.items():
        try:
            e0=torch.zeros(0,dtype=torch.long,device=dev)
            g=m.graph(torch.tensor(pos,dtype=torch.float32,device=dev).reshape(-1,3), torch.zeros(ns,3,3,device=dev), e0,e0,torch.zeros((0,3),dtype=torch.long,device=dev), torch.tensor(z,dtype=torch.long,device=dev), torch.tensor(sysidx,dtype=torch.int32,device=dev))
            a=m.forward(g); gr=m.backward(g, torch.ones_like(a))
            msg=f"atomic {tuple(a.shape)} grad {tuple(gr.shape)} finite {bool(torch.isfinite(a).all())}"
            if len(z):
                p64={k:v.double() for k,v in params.items()}
                e_ref,g_ref,a_ref=osoap.energy_and_gradient(p64,hy,types,torch.tensor(pos).double(),torch.zeros(ns,3,3,dtype=torch.float64),e0.cpu(),e0.cpu(),torch.zeros((0,3),dtype=torch.long),torch.tensor(z),torch.tensor(sysidx).long())
                msg+=f" err {float((a.cpu().double()-a_ref).abs().max()/a_ref.abs().max().clamp(min=1e-30)):.2e} aref {float(a_ref.abs().max()):.2e}"
                if legacy:
                    m.zero_grad(); a=m.forward(g); t=m.train_gradients(g, torch.ones_like(a), torch.zeros(len(z),3,device=dev)); msg+=" train OK"
            print(legacy,name,"OK",msg)
        except Exception as exc:
            print(legacy,name,"FAILED",type(exc).__name__,str(exc)[:300].replace("\n"," | "))
