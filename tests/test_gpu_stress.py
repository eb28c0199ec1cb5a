"""GPU robustness checks of the hand-overs the tuned path relies on (no reference counterpart: these guard the
implementation, not the arithmetic).

* node_split: the four-workgroups-per-row-tile node kernels (k_node2 / k_node_bwd2 SPLIT, graphs of at most 4 096 atoms) pass
  partial tiles between workgroups on different XCDs through device-coherent stores and loads without a cache write-back
  fence (csrc/pet_fwd.hip); a stale read would show as a run that differs from the first.
* the forward records on the graph, per workspace, whether it ran the fused attention block without saving Q, K, V; an
  adjoint configured for the three-kernel form must refuse that workspace."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TYPES = [1, 6, 7, 8]


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from metatrain_amd import runtime

    return runtime


def _setup(rt, n, seed):
    from metatrain_amd.pet import default_hypers
    from metatrain_amd.synthetic import random_box, synthetic_params

    dev = torch.device("cuda:0")
    hypers = default_hypers()
    model = rt.HipModel(hypers, TYPES)
    model.load({k: v.to(dev) for k, v in synthetic_params(hypers, TYPES, {"energy": 1}, 0).items()}, "energy")
    pos, z, cell = random_box(n, seed=seed)
    posd = pos.to(dev)
    pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"])
    g = rt.HipGraph(model, posd, cell.to(dev)[None], pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                    pairs[:, 2:5].contiguous(), z.to(dev), torch.zeros(n, dtype=torch.int32, device=dev))
    return model, g, torch.ones(n, device=dev)


@pytest.mark.parametrize("n", [33, 1000, 4096])
def test_node_split_hand_over_is_bit_stable(rt, n):
    model, g, ones = _setup(rt, n, seed=n)
    fw = rt.HipForward(model, g)
    a0 = fw.forward().clone()
    g0 = fw.backward(ones).clone()
    rt.config_set("node_split", 0)
    try:
        a1 = fw.forward().clone()
        g1 = fw.backward(ones).clone()
    finally:
        rt.config_set("node_split", 1)
    assert torch.equal(a0, a1), "the split forward is bit-identical to the unsplit kernel by construction"
    assert float((g0 - g1).abs().max() / g1.abs().max()) < 1e-5
    for _ in range(150):
        assert torch.equal(fw.forward(), a0)
        assert torch.equal(fw.backward(ones), g0)


def test_adjoint_refuses_a_workspace_whose_forward_did_not_save_qkv(rt):
    rt.config_set("attn_fused", 7)  # force the fused block on this small graph (the tile plan is made with the graph)
    try:
        model, g, ones = _setup(rt, 600, seed=5)
        fw = rt.HipForward(model, g)
        fw.forward()
        ref = fw.backward(ones).clone()
        fw.forward()
        rt.config_set("attn_fused", 0)  # the three-kernel adjoint would read a QKV nobody wrote
        with pytest.raises(rt.PetHipError):
            fw.backward(ones)
        rt.config_set("attn_fused", 7)
        fw.forward()
        assert torch.equal(fw.backward(ones), ref)
    finally:
        rt.config_set("attn_fused", 3)


@pytest.mark.parametrize("train", [False, True])
def test_large_batch_kernels_are_bit_stable(rt, train):
    """Run-to-run bit determinism of forward and adjoint on a batch that hands every stage to the large-graph kernels (32 x 1 000
    atoms: 600 k edge rows, 32 000 atoms: csrc/pet_emlp_s.hip, pet_head_s.hip, pet_compress_s.hip, pet_center_s.hip,
    pet_comb_s.hip). The first versions of k_comb_s and k_rowlin_s re-requested their row tile by LDS-DMA while the ds_read_b128 of
    its previous contents had been issued but not returned: an L2-warm DMA lands sooner than sixteen queued reads of a busy CU are
    served, and about one launch in seventy came out different (tools/debug/comb_det.py runs the longer version)."""
    from metatrain_amd import data
    from metatrain_amd.pet import default_hypers
    from metatrain_amd.synthetic import random_box, synthetic_params

    dev = torch.device("cuda:0")
    hypers = default_hypers()
    model = rt.HipModel(hypers, [1, 6, 7, 8])
    model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
    boxes = [random_box(1000, 300 + b) for b in range(32)]
    g = data.graph_of(model, data.collate([(p.to(dev), z.to(dev), c.to(dev), (True, True, True)) for p, z, c in boxes], 4.5))
    fw = rt.HipForward(model, g, train=train)
    a0 = fw.forward().clone()
    ones = torch.ones_like(a0)
    g0 = fw.backward(ones).clone()
    for _ in range(25):
        assert torch.equal(fw.forward(), a0)
        assert torch.equal(fw.backward(ones), g0)
