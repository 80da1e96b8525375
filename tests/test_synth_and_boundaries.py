"""CPU: the product-side synthetic-workload helpers, and the import boundary of oracle/ (test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may touch it)."""
import ast
import os

import torch

from oracle import nets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_init_state_is_deterministic_and_keeps_the_reference_layout():
    from tracknetv3_amd.utils import synth
    from tracknetv3_amd.utils.general import get_model
    a = synth.init_state_(get_model("TrackNet", 8, "concat"), 31).state_dict()
    b = synth.init_state_(get_model("TrackNet", 8, "concat"), 31).state_dict()
    c = synth.init_state_(get_model("TrackNet", 8, "concat"), 32).state_dict()
    shapes = nets.tracknet_state_shapes(27, 8)
    assert list(a) == list(shapes) and all(tuple(a[k].shape) == tuple(shapes[k][0]) and a[k].dtype == shapes[k][1] for k in a)
    assert all(torch.equal(a[k], b[k]) for k in a) and any(not torch.equal(a[k], c[k]) for k in a)
    assert all(bool(torch.isfinite(v.float()).all()) for v in a.values())
    w = a["down_block_1.conv_1.conv.weight"]
    bound = 2.4 / (27 * 9) ** 0.5
    assert w.abs().max().item() <= bound and w.abs().max().item() > 0.9 * bound
    assert 0.5 <= a["bottleneck.conv_2.bn.running_var"].min().item() and a["bottleneck.conv_2.bn.running_var"].max().item() <= 2.0
    fresh = synth.init_state_(get_model("TrackNet", 3, ""), 5, calibrated=False).state_dict()
    assert bool((fresh["down_block_1.conv_1.bn.weight"] == 1).all()) and bool((fresh["down_block_1.conv_1.bn.running_mean"] == 0).all())
    # the calibrated state keeps activations alive through the 17 layers (checked with the oracle on a small input)
    x = nets.synth_input((1, 27, 32, 64), 3)
    with torch.no_grad():
        p = nets.tracknet_forward({k: v.clone() for k, v in a.items()}, x, training=False)
    assert 1e-3 < p.std().item() and 0.0 < p.min().item() and p.max().item() < 1.0
    inp = synth.init_state_(get_model("InpaintNet"), 7).state_dict()
    assert list(inp) == list(nets.inpaintnet_state_shapes())


def test_disc_heatmaps_follow_the_dataset_format():
    from tracknetv3_amd.utils import synth
    y = synth.disc_heatmaps(4, 8, 72, 128, 11)
    assert y.shape == (4, 8, 72, 128) and y.dtype == torch.float32
    assert set(y.unique().tolist()) <= {0.0, 1.0}
    flat = y.view(32, -1).sum(1)
    assert all(flat[i] == 0 for i in range(32) if i % 5 == 4)          # every fifth map is empty
    assert flat.max().item() <= 21 and flat[[i for i in range(32) if i % 5 != 4]].min().item() >= 6   # a radius-2.5 disc, clipped at borders
    assert torch.equal(y, synth.disc_heatmaps(4, 8, 72, 128, 11))


def _oracle_imports(path):
    tree = ast.parse(open(path).read())
    hits = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
            hits.append(node)
        if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
            hits.append(node)
    return tree, hits


def test_oracle_is_only_imported_where_it_may_be():
    # the product package never imports it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tracknetv3_amd")):
        for f in files:
            if f.endswith(".py"):
                assert not _oracle_imports(os.path.join(dirpath, f))[1], f
    # bench.py: only inside cpu_baseline();  __graft_entry__.py: only inside smoke()
    for fname, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        tree, hits = _oracle_imports(os.path.join(ROOT, fname))
        assert hits, fname
        inside = set()
        for fn in ast.walk(tree):
            if isinstance(fn, ast.FunctionDef) and fn.name == allowed:
                inside = {id(n) for n in ast.walk(fn)}
        assert all(id(h) in inside for h in hits), fname
