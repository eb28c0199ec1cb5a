import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import pet as opet
from test_gpu_train import _inputs, _oracle_param_grads_cond, _oracle_param_grads
from metatrain_amd import runtime as rt
TYPES=[1,6,7,8]
dev=torch.device("cuda:0")
gd=os.path.join(os.getcwd(),"tests","golden")
base=dict(d_pet=8, d_head=8, d_node=8, d_feedforward=8, num_heads=1, num_attention_layers=1, num_gnn_layers=1)
for name, extra, cond in [("res nocond", dict(base, featurizer_type="residual"), False), ("res cond", dict(base, featurizer_type="residual"), True), ("ff cond", base, True),
                          ("res cond L2", dict(base, featurizer_type="residual", num_gnn_layers=2), True)]:
    hypers=dict(opet.DEFAULT_HYPERS, system_conditioning=cond, **extra)
    params=opet.synthetic_params(hypers, TYPES, {"energy":1}, 0, torch.float32)
    inp=_inputs(gd,"batch_two_systems.npz")
    kw={}
    if cond:
        inp["charge"], inp["spin_multiplicity"]=torch.tensor([-2,3]), torch.tensor([1,4]); kw=dict(charge=inp["charge"], spin_multiplicity=inp["spin_multiplicity"])
    n=inp["positions"].shape[0]
    model=rt.HipModel(hypers,TYPES); model.load({k:v.to(dev) for k,v in params.items()},"energy")
    graph=rt.HipGraph(model, inp["positions"].float().to(dev), inp["cells"].float().to(dev), inp["centers"].to(dev), inp["neighbors"].to(dev), inp["cell_shifts"].to(dev), inp["species"].to(dev), inp["system_indices"].int().to(dev))
    if cond: graph.set_conditioning(inp["charge"].to(dev), inp["spin_multiplicity"].to(dev), inp["system_indices"].to(dev))
    p64={k:(v if k=="species_to_species_index" else v.double()) for k,v in params.items()}
    a_ref=opet.pet_atomic_energies(p64,hypers,inp["positions"].double(),inp["cells"].double(),inp["centers"],inp["neighbors"],inp["cell_shifts"],inp["species"],inp["system_indices"].long(),"energy",**kw)[:,0]
    a_inf=rt.HipForward(model,graph).forward()
    fw=rt.HipForward(model,graph,train=True)
    a_tr=fw.forward()
    print(name,"inference err",float((a_inf.cpu().double()-a_ref).abs().max()/a_ref.abs().max()),"train-forward err",float((a_tr.cpu().double()-a_ref).abs().max()/a_ref.abs().max()))
    w=torch.ones(n)
    ref=(_oracle_param_grads_cond if cond else _oracle_param_grads)(params,hypers,inp,w)
    model.zero_grad(); fw.forward(); fw.backward_train(w.to(dev))
    got=model.grads()
    worst=sorted(((float((got[k].cpu().double()-r).abs().max()/max(1e-12,r.abs().max())),k) for k,r in ref.items()),reverse=True)[:4]
    print("   grads worst",worst)
