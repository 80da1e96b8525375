#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE.

Run in the survey/build container only (it needs /root/reference):

    python tests/golden/make_golden.py

It (1) pins the oracle (oracle/nets.py, oracle/postproc.py) against the imported
reference modules / live-extracted reference functions on identical inputs and
asserts equality, and (2) writes small .npz fixtures (inputs are regenerated from
the portable PRNG, so only expected OUTPUTS are stored).  Only data is written --
no reference source text is copied anywhere.

What is live here:
  * model.TrackNet / model.InpaintNet / utils.metric.WBCELoss  -- imported as modules
  * get_ensemble_weight, generate_inpaint_mask (test.py), mixup (train.py),
    get_model (utils/general.py), predict (predict.py), to_img/to_img_format and the
    two temporal-ensemble loops of predict.py's __main__ -- extracted with ``ast`` from
    the reference files and exec'd against stub namespaces (their modules cannot be
    imported: cv2 / pycocotools / tensorboard are absent)
  * predict_location is NOT live (OpenCV absent): parity unpinned at that call.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

import model as ref_model                      # noqa: E402  (reference)
from utils.metric import WBCELoss as ref_wbce   # noqa: E402  (reference)

from oracle import nets, postproc, prng         # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
torch.set_num_threads(8)


# --------------------------------------------------------------------- ast helpers
def _src(path):
    with open(os.path.join(REF, path)) as f:
        return f.read()


def extract_function(path, name, namespace):
    tree = ast.parse(_src(path))
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, f"<ref:{path}:{name}>", "exec"), namespace)
            return namespace[name]
    raise KeyError(name)


import math  # noqa: E402

ref_get_ensemble_weight = extract_function("test.py", "get_ensemble_weight", {"torch": torch, "math": math})
ref_generate_inpaint_mask = extract_function("test.py", "generate_inpaint_mask", {"np": np})
ref_get_model = extract_function("utils/general.py", "get_model",
                                 {"TrackNet": ref_model.TrackNet, "InpaintNet": ref_model.InpaintNet})
_general_ns = {"np": np, "HEIGHT": 288, "WIDTH": 512}
ref_to_img = extract_function("utils/general.py", "to_img", _general_ns)
ref_to_img_format = extract_function("utils/general.py", "to_img_format", _general_ns)
ref_mixup = extract_function("train.py", "mixup", {"np": np, "torch": torch})
ref_predict = extract_function(
    "predict.py", "predict",
    {"torch": torch, "np": np, "WIDTH": 512, "HEIGHT": 288, "to_img": ref_to_img,
     "to_img_format": ref_to_img_format, "predict_location": postproc.predict_location})


ref_evaluate = extract_function(
    "test.py", "evaluate",
    {"torch": torch, "np": np, "math": math, "WIDTH": 512, "HEIGHT": 288, "to_img": ref_to_img,
     "to_img_format": ref_to_img_format, "predict_location": postproc.predict_location,
     "pred_types_map": {t: i for i, t in enumerate(["TP", "TN", "FP1", "FP2", "FN"])}})
ref_get_metric = extract_function("utils/metric.py", "get_metric", {})


def extract_ensemble_loops():
    """Return (heat_stmts, coor_stmts): the statement lists of predict.py's two
    temporal-ensemble loops (predict.py:163-209 and predict.py:243-301)."""
    tree = ast.parse(_src("predict.py"))
    main_if = [n for n in tree.body if isinstance(n, ast.If)][-1]

    def is_nonoverlap_test(n):
        return isinstance(n, ast.If) and "nonoverlap" in ast.unparse(n.test)

    heat, coor = None, None
    for n in ast.walk(main_if):
        if is_nonoverlap_test(n):
            body = n.orelse
            # keep from the "Init ... buffer params" assignments on
            start = next(k for k, s in enumerate(body) if "sample_count" in ast.unparse(s))
            if "y_pred_buffer" in ast.unparse(ast.Module(body=body, type_ignores=[])):
                heat = body[start:]      # skip video decoding / dataset construction
            else:
                coor = body              # dataset / loader constructors are stubbed
    assert heat is not None and coor is not None
    return heat, coor


class _Dev:
    """Stub for tensors that get .float().cuda() / model outputs that get .detach().cpu()."""
    def __init__(self, t):
        self.t = t

    def float(self):
        return self            # keep the integer window ids intact

    def cuda(self):
        return self.t


def run_ref_heat_ensemble(windows, seq_len, eval_mode, batch, hw):
    """Drive the reference's own heat-map ensemble loop with precomputed window outputs."""
    heat_stmts, _ = extract_ensemble_loops()
    n_win = windows.shape[0]
    video_len = n_win + seq_len - 1
    loader, k = [], 0
    while k < n_win:
        b = min(batch, n_win - k)
        idx = torch.zeros((b, seq_len, 2), dtype=torch.int64)
        for j in range(b):
            idx[j, :, 1] = torch.arange(k + j, k + j + seq_len)
        loader.append((idx, _Dev(torch.arange(k, k + b))))
        k += b
    recorded = []

    def predict_stub(i, y_pred=None, c_pred=None, img_scaler=(1, 1)):
        recorded.append((i.clone(), (y_pred if y_pred is not None else c_pred).clone()))
        return {"Frame": [], "X": [], "Y": [], "Visibility": []}

    class _Out:
        def __init__(self, t):
            self.t = t

        def detach(self):
            return self

        def cpu(self):
            return self.t

    ns = {"torch": torch, "HEIGHT": hw[0], "WIDTH": hw[1], "video_len": video_len, "seq_len": seq_len,
          "args": types.SimpleNamespace(eval_mode=eval_mode), "get_ensemble_weight": ref_get_ensemble_weight,
          "data_loader": loader, "tqdm": lambda x: x, "img_scaler": (1, 1), "predict": predict_stub,
          "tracknet": lambda ids: _Out(torch.from_numpy(windows[ids.numpy()])),
          "tracknet_pred_dict": {"Frame": [], "X": [], "Y": [], "Visibility": []}}
    exec(compile(ast.Module(body=heat_stmts, type_ignores=[]), "<ref:predict.py:heat-ensemble>", "exec"), ns)
    frames = torch.cat([r[0] for r in recorded], 0)[:, 0, 1].numpy()
    ens = torch.cat([r[1] for r in recorded], 0)[:, 0].numpy()
    return frames, ens


def run_ref_coor_ensemble(windows, masks, coor_in, seq_len, eval_mode, batch):
    """Drive the reference's coordinate-ensemble loop; ``windows`` are InpaintNet outputs."""
    _, coor_stmts = extract_ensemble_loops()
    n_win = windows.shape[0]
    loader, k = [], 0
    while k < n_win:
        b = min(batch, n_win - k)
        idx = torch.zeros((b, seq_len, 2), dtype=torch.int64)
        for j in range(b):
            idx[j, :, 1] = torch.arange(k + j, k + j + seq_len)
        loader.append((idx, torch.from_numpy(coor_in[k:k + b]), torch.from_numpy(masks[k:k + b])))
        k += b
    recorded = []

    def predict_stub(i, y_pred=None, c_pred=None, img_scaler=(1, 1)):
        recorded.append((i.clone(), c_pred.clone()))
        return {"Frame": [], "X": [], "Y": [], "Visibility": []}

    class _Out:
        def __init__(self, t):
            self.t = t

        def detach(self):
            return self

        def cpu(self):
            return self.t

    class _DS(list):
        pass

    counter = {"k": 0}

    def inpaintnet(c, m):
        b = c.shape[0]
        o = torch.from_numpy(windows[counter["k"]:counter["k"] + b])
        counter["k"] += b
        return _Out(o)

    class _T(torch.Tensor):
        pass

    # .cuda() on CPU tensors would fail: patch via wrapper namespace
    ds = _DS(range(n_win))
    ns = {"torch": torch, "seq_len": seq_len, "args": types.SimpleNamespace(eval_mode=eval_mode, batch_size=batch),
          "get_ensemble_weight": ref_get_ensemble_weight, "tqdm": lambda x: x, "img_scaler": (1, 1),
          "predict": predict_stub, "inpaintnet": inpaintnet, "COOR_TH": postproc.COOR_TH,
          "inpaint_pred_dict": {"Frame": [], "X": [], "Y": [], "Visibility": []},
          "Shuttlecock_Trajectory_Dataset": lambda **kw: ds, "tracknet_pred_dict": {},
          "DataLoader": lambda *a, **kw: loader, "num_workers": 0}
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        exec(compile(ast.Module(body=coor_stmts, type_ignores=[]), "<ref:predict.py:coor-ensemble>", "exec"), ns)
    finally:
        torch.Tensor.cuda = orig_cuda
    frames = torch.cat([r[0] for r in recorded], 0)[:, 0, 1].numpy()
    ens = torch.cat([r[1] for r in recorded], 0)[:, 0].numpy()
    return frames, ens


# --------------------------------------------------------------------- network goldens
def load_ref_tracknet(in_dim, out_dim, sd):
    m = ref_model.TrackNet(in_dim, out_dim)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


def probes(t, n, seed):
    """n pseudo-random flat indices + values."""
    flat = t.reshape(-1)
    idx = (prng.uniform((n,), seed).astype(np.float64) * flat.shape[0]).astype(np.int64)
    return idx, flat[idx].copy()


def tracknet_case(tag, in_dim, out_dim, n, h, w, seed, calibrated, store_full):
    shapes = nets.tracknet_state_shapes(in_dim, out_dim)
    sd = nets.synth_state(shapes, seed, calibrated=calibrated)
    ref = load_ref_tracknet(in_dim, out_dim, sd)
    assert list(ref.state_dict().keys()) == list(shapes.keys()), "state_dict key order differs"
    for k, v in ref.state_dict().items():
        assert tuple(v.shape) == tuple(shapes[k][0]) and v.dtype == shapes[k][1], k
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)
    y = nets.disc_heatmaps(n, out_dim, h, w, seed + 2000)
    out = {}

    # ---- eval forward: reference vs oracle (same torch build -> expect tiny/zero diff)
    ref.eval()
    with torch.no_grad():
        p_ref = ref(x)
        p_orc = nets.tracknet_forward(sd, x, training=False)
        p_orc64 = nets.tracknet_forward(sd, x.double(), training=False)
    d = (p_ref - p_orc).abs().max().item()
    d64 = (p_ref.double() - p_orc64).abs().max().item()
    print(f"[{tag}] eval  ref-vs-oracle32 {d:.3e}  ref-vs-oracle64 {d64:.3e}  range [{p_ref.min():.4f},{p_ref.max():.4f}]")
    assert d <= 2e-6 and d64 <= 2e-5
    out["eval_loss"] = np.float64(ref_wbce(p_ref, y).item())
    assert abs(nets.wbce_loss(p_ref, y).item() - out["eval_loss"]) <= 1e-7

    # ---- train-mode forward + WBCE + backward on the reference
    ref.train()
    ref.zero_grad()
    p_tr = ref(x)
    loss = ref_wbce(p_tr, y)
    loss.backward()
    sd_after = ref.state_dict()
    l_o, p_o, g_o, st_o = nets.tracknet_train_step_grads(sd, x, y, torch.float32)
    l_64, p_64, g_64, st_64 = nets.tracknet_train_step_grads(sd, x, y, torch.float64)
    dtr = (p_tr.detach() - p_o).abs().max().item()
    dtr64 = (p_tr.detach().double() - p_64).abs().max().item()
    print(f"[{tag}] train ref-vs-oracle32 {dtr:.3e}  ref-vs-oracle64 {dtr64:.3e}  loss {loss.item():.6f}")
    assert dtr <= 5e-5 and dtr64 <= 5e-4
    assert abs(l_o.item() - loss.item()) <= 1e-6
    for k, v in st_o.items():
        rv = sd_after[k]
        assert torch.allclose(v.to(rv.dtype), rv, rtol=1e-4, atol=1e-6), k
    gstat_names, gstats = [], []
    worst = 0.0
    for name, prm in ref.named_parameters():
        g_ref = prm.grad.detach().double()
        gm = g_64[name].abs().max().item() + 1e-30
        worst = max(worst, (g_ref - g_64[name]).abs().max().item() / gm)
        gi, gv = probes(g_64[name].numpy(), 32, prng.name_seed(name, 7) % (1 << 31))
        gstat_names.append(name)
        gstats.append(np.concatenate([[g_64[name].sum().item(), g_64[name].abs().sum().item(),
                                       g_64[name].abs().max().item()], gv]))
        out.setdefault("grad_probe_idx", {})[name] = gi
    print(f"[{tag}] grads ref32-vs-oracle64 worst rel-to-max {worst:.3e}")
    out["train_loss"] = np.float64(loss.item())
    out["train_loss64"] = np.float64(l_64.item())
    out["grad_names"] = np.array(gstat_names)
    out["grad_stats64"] = np.stack(gstats)            # [sum, abs-sum, abs-max, 32 probes] per param (fp64 oracle)
    out["grad_probe_idx"] = np.stack([out["grad_probe_idx"][k] for k in gstat_names])
    out["grad_ref32_vs_64_worst"] = np.float64(worst)
    bn_names = [k for k in sd_after if "running_" in k]
    out["bn_names"] = np.array(bn_names)
    out["bn_after"] = np.concatenate([sd_after[k].numpy().ravel() for k in bn_names])

    if store_full:
        out["eval_out"] = p_ref.numpy()
        out["train_out"] = p_tr.detach().numpy()
    pi, pv = probes(p_ref.numpy(), 4096, seed + 5)
    out["eval_probe_idx"], out["eval_probe_val"] = pi, pv
    out["eval_probe_val64"] = p_orc64.numpy().reshape(-1)[pi]
    out["eval_chan_sum64"] = p_orc64.sum(dim=(2, 3)).numpy()
    pi, pv = probes(p_tr.detach().numpy(), 4096, seed + 6)
    out["train_probe_idx"], out["train_probe_val"] = pi, pv
    out["meta"] = np.array([in_dim, out_dim, n, h, w, seed, int(calibrated)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"tracknet_{tag}.npz"), **out)


def tracknet_fullsize_case():
    """(27,8) at 288x512, N=1, calibrated weights, eval: probes + channel sums only."""
    in_dim, out_dim, n, h, w, seed = 27, 8, 1, 288, 512, 31
    shapes = nets.tracknet_state_shapes(in_dim, out_dim)
    sd = nets.synth_state(shapes, seed, calibrated=True)
    ref = load_ref_tracknet(in_dim, out_dim, sd).eval()
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)
    with torch.no_grad():
        p_ref = ref(x)
        taps = {}
        p_orc = nets.tracknet_forward(sd, x, training=False, taps=taps)
    d = (p_ref - p_orc).abs().max().item()
    print(f"[full] eval ref-vs-oracle32 {d:.3e} range [{p_ref.min():.4f},{p_ref.max():.4f}] mean {p_ref.mean():.4f}")
    assert d <= 5e-6
    pi, pv = probes(p_ref.numpy(), 8192, seed + 5)
    out = {"eval_probe_idx": pi, "eval_probe_val": pv, "eval_chan_sum": p_ref.double().sum(dim=(2, 3)).numpy(),
           "eval_min": np.float32(p_ref.min().item()), "eval_max": np.float32(p_ref.max().item()),
           "meta": np.array([in_dim, out_dim, n, h, w, seed, 1], dtype=np.int64)}
    # per-layer activation checksums (helps localise a failing kernel on the GPU box)
    out["tap_names"] = np.array(list(taps.keys()))
    out["tap_abs_mean"] = np.array([taps[k].abs().mean().item() for k in taps])
    np.savez_compressed(os.path.join(OUT, "tracknet_27_8_288x512.npz"), **out)


def wbce_case():
    p = torch.tensor([0.0, 4e-8, 1e-7, 3e-7, 0.25, 0.5, 0.75, 1 - 6e-8, 1 - 1.2e-7, 1.0], dtype=torch.float32)
    ys = torch.tensor([0.0, 1.0, 0.3], dtype=torch.float32)
    P = p[None, :].repeat(3, 1).reshape(1, 1, 3, 10).clone().requires_grad_(True)
    Y = ys[:, None].repeat(1, 10).reshape(1, 1, 3, 10)
    loss = ref_wbce(P, Y)
    loss.backward()
    g_cf = nets.wbce_grad_closed_form(P.detach(), Y)
    assert torch.allclose(P.grad, g_cf, rtol=1e-5, atol=1e-9), (P.grad - g_cf).abs().max()
    per = ref_wbce(P.detach(), Y, reduce=False)
    assert per.shape == (1,)
    np.savez(os.path.join(OUT, "wbce_edge.npz"), p=P.detach().numpy(), y=Y.numpy(), loss=np.float64(loss.item()),
             grad=P.grad.numpy(), per_sample=per.numpy())
    print(f"[wbce] edge loss {loss.item():.6f}")


def inpaint_case():
    shapes = nets.inpaintnet_state_shapes()
    sd = nets.synth_state(shapes, 77)
    ref = ref_model.InpaintNet()
    assert list(ref.state_dict().keys()) == list(shapes.keys())
    ref.load_state_dict(sd, strict=True)
    n, L = 6, 16
    coor = nets.synth_input((n, L, 2), 501)
    vis = (nets.synth_input((n, L, 1), 502) > 0.2).float()
    coor = coor * vis
    mask = ((nets.synth_input((n, L, 1), 503) < 0.3).float() * vis)
    gt = nets.synth_input((n, L, 2), 504)
    ref.train()
    cin = coor * (1 - mask)
    o = ref(cin, mask)
    loss = torch.nn.MSELoss()(o * mask, gt * mask)
    loss.backward()
    with torch.no_grad():
        o2 = nets.inpaintnet_forward(sd, cin, mask)
    d = (o.detach() - o2).abs().max().item()
    print(f"[inpaint] ref-vs-oracle {d:.3e} loss {loss.item():.6f}")
    assert d <= 1e-6
    assert abs(nets.inpaint_masked_mse(o.detach(), gt, mask).item() - loss.item()) < 1e-8
    names = [k for k, _ in ref.named_parameters()]
    np.savez_compressed(os.path.join(OUT, "inpaintnet_6x16.npz"), out=o.detach().numpy(), loss=np.float64(loss.item()),
                        grad_names=np.array(names),
                        grad_sums=np.array([p.grad.double().sum().item() for _, p in ref.named_parameters()]),
                        grad_abs=np.array([p.grad.double().abs().sum().item() for _, p in ref.named_parameters()]),
                        grad_pred_w=ref.predictor.weight.grad.numpy(), grad_down1_w=ref.down_1.conv.weight.grad.numpy())


def host_logic_cases():
    out = {}
    # get_model channel plan + error behaviour
    plan = []
    for bg in ("", "subtract", "subtract_concat", "concat", None, "zzz"):
        for L in (1, 3, 8):
            m = ref_get_model("TrackNet", L, bg)
            cin = m.down_block_1.conv_1.conv.weight.shape[1]
            cout = m.predictor.weight.shape[0]
            assert (cin, cout) == nets.tracknet_dims(L, bg), (bg, L)
            plan.append((L, cin, cout))
    try:
        ref_get_model("Nope")
        raise AssertionError
    except ValueError as e:
        assert str(e) == "Invalid model name."
    assert isinstance(ref_get_model("InpaintNet"), ref_model.InpaintNet)
    out["get_model_plan"] = np.array(plan)
    # ensemble weights
    for L in (1, 2, 3, 5, 8, 16):
        for mode in ("average", "weight"):
            a = ref_get_ensemble_weight(L, mode).numpy()
            b = postproc.get_ensemble_weight(L, mode)
            assert np.array_equal(a, b), (L, mode, a, b)
            out[f"ens_w_{mode}_{L}"] = a
    # generate_inpaint_mask on hand cases + random cases
    cases = [
        ([0, 0, 1, 1, 1], [0, 0, 50, 60, 70]),
        ([1, 1, 0, 0, 1, 1], [40, 45, 0, 0, 50, 55]),
        ([1, 0, 0, 1, 1], [40, 0, 0, 50, 55]),          # gap starting at index 1: excluded by i > 1
        ([1, 1, 0, 0, 1], [40, 10, 0, 0, 55]),          # low neighbour
        ([1, 1, 1, 0, 0], [40, 41, 42, 0, 0]),          # trailing gap
        ([0, 0, 0], [0, 0, 0]),
        ([1, 1, 1], [5, 6, 7]),
    ]
    r = prng.uniform((40, 64), 9)
    for k in range(40):
        vis = (r[k] > 0.35).astype(int)
        yy = (prng.uniform((64,), 100 + k) * 120).astype(int) * vis
        cases.append((vis.tolist(), yy.tolist()))
    masks = []
    for vis, yy in cases:
        pd_ = {"Y": yy, "Visibility": vis}
        for th in (30, 14.4):
            a = ref_generate_inpaint_mask(pd_, th_h=th)
            b = postproc.generate_inpaint_mask(pd_, th_h=th)
            assert a == b, (vis, yy, th)
            masks.append(np.array(a))
    out["inpaint_mask_vis"] = np.array([np.pad(np.array(c[0]), (0, 64 - len(c[0])), constant_values=-1) for c in cases])
    out["inpaint_mask_y"] = np.array([np.pad(np.array(c[1]), (0, 64 - len(c[1])), constant_values=-1) for c in cases])
    out["inpaint_mask_out"] = np.array([np.pad(m, (0, 64 - len(m)), constant_values=-1) for m in masks])
    # mixup with injected lambda / perm (patch RNGs)
    x = nets.synth_input((4, 3, 8, 16), 11)
    y = nets.synth_input((4, 2, 8, 16), 12)
    lam = np.array([0.2, 0.9, 0.5, 0.61])
    perm = torch.tensor([2, 0, 3, 1])
    o_beta, o_perm = np.random.beta, torch.randperm
    np.random.beta = lambda a, b, size=None: lam.copy()
    torch.randperm = lambda n: perm.clone()
    try:
        xm, ym = ref_mixup(x, y, 0.5)
    finally:
        np.random.beta, torch.randperm = o_beta, o_perm
    xo, yo = nets.mixup_injected(x, y, lam, perm)
    assert torch.equal(xm, xo) and torch.equal(ym, yo)
    out["mixup_x"], out["mixup_y"], out["mixup_lam"], out["mixup_perm"] = xm.numpy(), ym.numpy(), lam, perm.numpy()
    # predict(): coordinate path + heat-map path (peak-find = restated, see header)
    idx = np.zeros((3, 4, 2), dtype=np.int64)
    idx[:, :, 1] = np.array([[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 9, 9]])     # padded tail de-dup
    c = prng.uniform((3, 4, 2), 21)
    c[0, 1] = 0
    # predict.py:51 `int(c_p[0] * WIDTH * img_scaler[0])`: c_p[0] is an np.float32 SCALAR and WIDTH a Python int.  Under the
    # reference's pinned numpy 1.22.4 (requirements.txt:2; legacy promotion: float32 scalar x Python scalar -> float64) the
    # product is float64; under this container's numpy 2.2 (NEP 50) it would stay float32 and truncate differently for 45 % of
    # the coordinates X/1920.  The pinned environment is the reference: its own `predict` is therefore run here on the fp32
    # values widened to float64 (an exact conversion), which makes numpy 2.2 evaluate exactly numpy 1.22.4's expression.
    a = ref_predict(torch.from_numpy(idx), c_pred=torch.from_numpy(c).double(), img_scaler=(3.75, 3.75))
    b = postproc.predict(idx, c_pred=c, img_scaler=(3.75, 3.75))
    assert a == b
    out["predict_c_idx"], out["predict_c_in"] = idx, c
    out["predict_c_out"] = np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]])
    # ... and on coordinates of the form fp32(X / 1920), fp32(Y / 1080) -- what an unmasked frame carries through the
    # InpaintNet stage -- where float32 and float64 arithmetic disagree about int() most often
    n_c = 240
    idx2 = np.zeros((n_c, 1, 2), dtype=np.int64)
    idx2[:, 0, 1] = np.arange(n_c)
    c2 = np.zeros((n_c, 1, 2), dtype=np.float32)
    c2[:, 0, 0] = (np.arange(n_c) * 8 + 3).astype(np.float64) / 1920
    c2[:, 0, 1] = (np.arange(n_c) * 4 + 1).astype(np.float64) / 1080
    a = ref_predict(torch.from_numpy(idx2), c_pred=torch.from_numpy(c2).double(), img_scaler=(3.75, 3.75))
    b = postproc.predict(idx2, c_pred=c2, img_scaler=(3.75, 3.75))
    assert a == b
    a32 = ref_predict(torch.from_numpy(idx2), c_pred=torch.from_numpy(c2), img_scaler=(3.75, 3.75))
    assert a32 != a, "expected numpy-2 float32 promotion to differ from the pinned numpy's float64 on these inputs"
    out["predict_c64_out"] = np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]])
    hm = np.zeros((3, 4, 288, 512), dtype=np.float32)
    hm[0, 0, 100:105, 200:207] = 0.9
    hm[0, 1, 10:12, 10:12] = 0.7
    hm[0, 1, 50:53, 300:303] = 0.8
    hm[1, 2, 0:3, 0:2] = 0.51
    hm[2, 0, 287, 511] = 1.0
    a = ref_predict(torch.from_numpy(idx), y_pred=torch.from_numpy(hm), img_scaler=(3.75, 3.75))
    b = postproc.predict(idx, y_pred=hm, img_scaler=(3.75, 3.75))
    assert a == b
    out["predict_h_out"] = np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]])
    np.savez_compressed(os.path.join(OUT, "host_logic.npz"), **out)
    print("[host] get_model / ensemble weights / inpaint mask / mixup / predict pinned")


from pipeline_common import evaluate_inputs   # noqa: E402  (deterministic inputs shared with the tests)


def _pack_eval(d):
    keys = ["Frame", "X", "Y", "Visibility", "Type"]
    out = {k: np.array(d[k]) for k in keys}
    for k in ("BBox", "Confidence", "X_GT", "Y_GT", "Visibility_GT"):
        if k in d:
            out[k] = np.array(d[k])
    return out


def evaluate_cases():
    idx, y_true, y_pred, c_true, c_pred = evaluate_inputs()
    out = {}
    for name, kw in (("h_plain", dict(tolerance=4.)), ("h_full", dict(tolerance=4., img_scaler=(3.75, 3.75), output_bbox=True, output_gt=True)),
                     ("h_tol1", dict(tolerance=1.))):
        a = ref_evaluate(torch.from_numpy(idx), y_true=torch.from_numpy(y_true.copy()), y_pred=torch.from_numpy(y_pred.copy()), **kw)
        b = postproc.evaluate(idx, y_true=y_true, y_pred=y_pred, **kw)
        assert a == b, name
        for k, v in _pack_eval(a).items():
            out[f"{name}_{k}"] = v
    for name, kw in (("c_plain", dict(tolerance=4.)), ("c_gt", dict(tolerance=4., img_scaler=(3.75, 3.75), output_gt=True))):
        a = ref_evaluate(torch.from_numpy(idx), c_true=torch.from_numpy(c_true.copy()), c_pred=torch.from_numpy(c_pred.copy()), **kw)
        b = postproc.evaluate(idx, c_true=c_true, c_pred=c_pred, **kw)
        assert a == b, name
        for k, v in _pack_eval(a).items():
            out[f"{name}_{k}"] = v
    rows = []
    for tp, tn, fp1, fp2, fn in ((0, 0, 0, 0, 0), (10, 5, 1, 2, 3), (0, 7, 0, 0, 0), (3, 0, 0, 0, 9), (0, 0, 4, 4, 0), (977, 13, 8, 14, 6)):
        a = ref_get_metric(tp, tn, fp1, fp2, fn)
        assert a == postproc.get_metric(tp, tn, fp1, fp2, fn)
        rows.append([tp, tn, fp1, fp2, fn, *a])
    out["get_metric"] = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "evaluate.npz"), **out)
    print("[host] evaluate / get_metric pinned:", {k: v.tolist() for k, v in out.items() if k.endswith("_Type")})


def ensemble_cases():
    out = {}
    k = 0
    for L in (3, 8):
        for mode in ("weight", "average"):
            for n_win, batch in ((2, 4), (L - 1, 2), (L, 3), (L + 5, 4), (19, 5), (1, 1)):
                if n_win < 1:
                    continue
                hw = (4, 8)
                win = prng.uniform((n_win, L) + hw, 1000 + k)
                frames, ens = run_ref_heat_ensemble(win, L, mode, batch, hw)
                mine = np.concatenate(list(postproc.ensemble_stream(
                    [win[s:s + batch] for s in range(0, n_win, batch)], L, mode, n_win)), 0)
                assert ens.shape == mine.shape, (ens.shape, mine.shape)
                assert np.array_equal(frames, np.arange(n_win + L - 1))
                d = np.abs(ens - mine).max()
                assert np.array_equal(ens, mine), (L, mode, n_win, batch, d)      # same summation order: bit-equal
                out[f"heat_{k}_meta"] = np.array([L, mode == "weight", n_win, batch, 1000 + k])
                out[f"heat_{k}_ens"] = ens
                k += 1
    # coordinate ensemble (InpaintNet stage), L=16
    L = 16
    for j, (n_win, batch, mode) in enumerate(((20, 6, "weight"), (16, 16, "average"), (5, 2, "weight"))):
        win = prng.uniform((n_win, L, 2), 3000 + j)
        cin = prng.uniform((n_win, L, 2), 3100 + j)
        cin[prng.uniform((n_win, L), 3150 + j) < 0.2] = 0
        msk = (prng.uniform((n_win, L, 1), 3200 + j) < 0.4).astype(np.float32)
        frames, ens = run_ref_coor_ensemble(win, msk, cin, L, mode, batch)
        blended = postproc.inpaint_blend_threshold(win, cin, msk)
        mine = np.concatenate(list(postproc.ensemble_stream(
            [blended[s:s + batch] for s in range(0, n_win, batch)], L, mode, n_win)), 0)
        th = (mine[:, 0] < postproc.COOR_TH) & (mine[:, 1] < postproc.COOR_TH)
        mine[th] = 0
        d = np.abs(ens - mine).max()
        assert np.array_equal(ens, mine), (n_win, batch, mode, d)             # torch's four-partial-sum order: bit-equal
        out[f"coor_{j}_meta"] = np.array([L, mode == "weight", n_win, batch, 3000 + j])
        out[f"coor_{j}_ens"] = ens
    np.savez_compressed(os.path.join(OUT, "ensemble.npz"), **out)
    print(f"[ensemble] {k} heat-map + 3 coordinate cases: reference loop == restatement")


if __name__ == "__main__":
    host_logic_cases()
    evaluate_cases()
    ensemble_cases()
    wbce_case()
    inpaint_case()
    tracknet_case("9_3_32x64", 9, 3, 2, 32, 64, 13, calibrated=False, store_full=True)
    tracknet_case("27_8_32x64_cal", 27, 8, 2, 32, 64, 17, calibrated=True, store_full=True)
    tracknet_case("9_3_64x128_cal", 9, 3, 1, 64, 128, 19, calibrated=True, store_full=False)
    tracknet_fullsize_case()
    print("golden vectors written to", OUT)
