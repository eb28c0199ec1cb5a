import sys, os, traceback
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import pet as opet
from metatrain_amd.pet import PETBackend
TYPES=[1,6,7,8]
dev=torch.device("cuda:0")
g=dict(np.load("tests/golden/batch_two_systems.npz")); t=lambda k: torch.tensor(g[k]).to(dev)
hypers=dict(opet.DEFAULT_HYPERS)
params=opet.synthetic_params(hypers,TYPES,{"energy":1},0,torch.float32)
def run(be):
    pos=t("in_positions").float().requires_grad_(True); cells=t("in_cells").float()
    batch=be.preprocess(pos,t("in_centers"),t("in_neighbors"),t("in_species"),cells,t("in_cell_shifts"),t("in_system_indices"),1.0)
    nodes,edges=be.calculate_features(batch)
    pred,_,_=be.predict(nodes,edges,batch,cells,t("in_system_indices"),["energy"])
    e=pred["energy"][0].sum(); (gr,)=torch.autograd.grad(e,pos)
    return float(e), gr
be=PETBackend(hypers,TYPES); be.add_output("energy",{"energy":[1]}); be.load_state_dict(params,strict=True); be=be.to(dev).eval()
e0,g0=run(be); print("eager",e0)
for fullgraph in (False, True):
    be2=PETBackend(hypers,TYPES); be2.add_output("energy",{"energy":[1]}); be2.load_state_dict(params,strict=True); be2=be2.to(dev).eval()
    try:
        with torch._dynamo.config.patch(capture_scalar_outputs=True, capture_dynamic_output_shape_ops=True, specialize_int=True):
            be2.preprocess=torch.compile(be2.preprocess,fullgraph=fullgraph)
            be2.calculate_features=torch.compile(be2.calculate_features,fullgraph=fullgraph)
            be2.predict=torch.compile(be2.predict,fullgraph=fullgraph)
            e1,g1=run(be2)
        print("compiled fullgraph",fullgraph,e1,float((g1-g0).abs().max()))
    except Exception as exc:
        print("compiled fullgraph",fullgraph,"FAILED:",type(exc).__name__,str(exc)[:600].replace("\n"," | "))
