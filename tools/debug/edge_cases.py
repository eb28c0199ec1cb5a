import sys, os, traceback
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import pet as opet, nl as onl
from metatrain_amd.pet import PETBackend
TYPES=[1,6,7,8]
dev=torch.device("cuda:0")
variants={"default":{}, "small":dict(d_pet=8,d_head=8,d_node=16,d_feedforward=8,num_heads=2), "adaptive_solver":dict(num_neighbors_adaptive=5.0),
          "adaptive_legacy":dict(num_neighbors_adaptive=5.0, adaptive_cutoff_method="grid") , "small_adaptive":dict(d_pet=8,d_head=8,d_node=16,d_feedforward=8,num_heads=2,num_neighbors_adaptive=5.0)}
systems={
 "empty": (np.zeros((0,3)), [], np.zeros((3,3)), [False]*3),
 "isolated": (np.zeros((1,3)), [6], np.zeros((3,3)), [False]*3),
 "dissociated": (np.array([[0,0,0],[0,0,100.0]]), [6,6], np.zeros((3,3)), [False]*3),
 "pair": (np.array([[0,0,0],[0,0,1.2]]), [6,8], np.zeros((3,3)), [False]*3),
}
for vn, extra in variants.items():
    hypers=dict(opet.DEFAULT_HYPERS, **extra)
    if "adaptive_cutoff_method" in extra and "adaptive_cutoff_method" not in opet.DEFAULT_HYPERS: pass
    params=opet.synthetic_params(hypers,TYPES,{"energy":1},0,torch.float32)
    be=PETBackend(hypers,TYPES); be.add_output("energy",{"energy":[1]}); be.load_state_dict(params,strict=True); be=be.to(dev).eval()
    p64={k:(v if k=="species_to_species_index" else v.double()) for k,v in params.items()}
    for sn,(pos,z,cell,pbc) in systems.items():
        try:
            i,j,s,_=onl.neighbor_list(pos,cell,pbc,hypers["cutoff"])
            P=torch.tensor(pos,dtype=torch.float32,device=dev).reshape(-1,3).requires_grad_(True)
            cells=torch.tensor(cell,dtype=torch.float32,device=dev)[None]
            ti=lambda a,dt=torch.long: torch.tensor(np.asarray(a),dtype=dt,device=dev)
            sysidx=torch.zeros(len(z),dtype=torch.long,device=dev)
            batch=be.preprocess(P,ti(i),ti(j),ti(z),cells,ti(s).reshape(-1,3),sysidx,float(hypers["cutoff_width_adaptive"]))
            nodes,edges=be.calculate_features(batch)
            pred,_,_=be.predict(nodes,edges,batch,cells,sysidx,["energy"])
            a=pred["energy"][0]
            msg=f"shape {tuple(a.shape)} finite {bool(torch.isfinite(a).all())}"
            (gr,)=torch.autograd.grad(a.sum(),P)
            msg+=f" grad shape {tuple(gr.shape)}"
            if len(z):
                ref=opet.pet_atomic_energies(p64,hypers,torch.tensor(pos).double().reshape(-1,3),torch.tensor(cell).double()[None],torch.tensor(i).long(),torch.tensor(j).long(),torch.tensor(s).long().reshape(-1,3),torch.tensor(z),torch.zeros(len(z),dtype=torch.long))
                msg+=f" err {float((a.cpu().double()-ref).abs().max()/ref.abs().max()):.2e} grad finite {bool(torch.isfinite(gr).all())}"
            print(vn,sn,"OK",msg)
        except Exception as exc:
            print(vn,sn,"FAILED",type(exc).__name__,str(exc)[:300].replace("\n"," | "))
