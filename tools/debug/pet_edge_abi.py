import sys, os, traceback
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import synthetic_params, random_box
dev=torch.device("cuda:0")
hy=default_hypers()
model=rt.HipModel(hy,[1,6,7,8]); model.load({k:v.to(dev) for k,v in synthetic_params(hy,[1,6,7,8],{"energy":1},0).items()},"energy")
def attempt(name, fn):
    try:
        print(name, "OK", fn())
    except Exception as exc:
        print(name, "FAILED", type(exc).__name__, str(exc)[:300].replace("\n"," | "))
cell=torch.eye(3)*10
attempt("nl empty", lambda: tuple(rt.neighbor_list(torch.zeros((0,3),device=dev), cell, [True]*3, hy["cutoff"])[0].shape))
attempt("nl one atom", lambda: tuple(rt.neighbor_list(torch.zeros((1,3),device=dev), cell, [True]*3, hy["cutoff"])[0].shape))
attempt("nl one atom small cell (self images)", lambda: tuple(rt.neighbor_list(torch.zeros((1,3),device=dev), torch.eye(3)*3.0, [True]*3, hy["cutoff"])[0].shape))
e0=torch.zeros(0,dtype=torch.int32,device=dev)
def empty_graph():
    g=rt.HipGraph(model, torch.zeros((0,3),device=dev), torch.zeros(1,3,3,device=dev), e0,e0,torch.zeros((0,3),dtype=torch.int32,device=dev), e0, e0)
    fw=rt.HipForward(model,g); a=fw.forward(); gr=fw.backward(torch.ones_like(a)); e=fw.sum_over_atoms(a)
    return tuple(a.shape), tuple(gr.shape), e.tolist()
attempt("empty graph forward/backward", empty_graph)
def empty_train():
    g=rt.HipGraph(model, torch.zeros((0,3),device=dev), torch.zeros(1,3,3,device=dev), e0,e0,torch.zeros((0,3),dtype=torch.int32,device=dev), e0, e0)
    fw=rt.HipForward(model,g,train=True); model.zero_grad(); a=fw.forward(); fw.backward_train(torch.ones_like(a)); return "grads finite %s" % bool(torch.isfinite(model.flat_grad()).all())
attempt("empty graph train", empty_train)
def gap_batch():
    pos,z,c=random_box(40,seed=1)
    pairs,_=rt.neighbor_list(pos.to(dev), c, [True]*3, hy["cutoff"])
    sysidx=torch.cat([torch.zeros(20),torch.full((20,),2)]).int().to(dev)   # system 1 is empty
    # pairs across the two halves are not physical here, keep only within-half pairs
    keep=(pairs[:,0]<20)==(pairs[:,1]<20); pairs=pairs[keep]
    g=rt.HipGraph(model,pos.to(dev),torch.stack([c,c,c]).to(dev),pairs[:,0].contiguous(),pairs[:,1].contiguous(),pairs[:,2:5].contiguous(),z.to(dev),sysidx)
    fw=rt.HipForward(model,g); a=fw.forward(); e=fw.sum_over_atoms(a); gr,gc=fw.backward(torch.ones_like(a),want_cell_grad=True)
    return [round(x,4) for x in e.tolist()], float(gc[1].abs().max())
attempt("batch with an empty system in the middle", gap_batch)
def batched_nl():
    boxes=[random_box(30,seed=s) for s in (1,2)]
    pos=[b[0].to(dev) for b in boxes]; pos.insert(1, torch.zeros((0,3),device=dev))
    cells=[boxes[0][2], torch.eye(3)*5, boxes[1][2]]
    pairs,_=rt.neighbor_list_batch(torch.cat(pos), torch.stack(cells), [[True]*3]*3, [0,30,30,60], hy["cutoff"])
    single=sum(rt.neighbor_list(p, c, [True]*3, hy["cutoff"])[0].shape[0] for p,c in ((pos[0],cells[0]),(pos[2],cells[2])))
    e,_=rt.neighbor_list_batch(torch.zeros((0,3),device=dev), torch.eye(3)[None], [[True]*3], [0,0], hy["cutoff"])
    return pairs.shape[0], single, tuple(e.shape)
attempt("batched nl with an empty system", batched_nl)
