"""Shared by the CPU (emulator) and GPU pipeline tests: a synthetic 'TrackNet' stub that paints clean heat maps from
frame ids carried in the frames, and the oracle-side restatement of predict.py's flow (numpy / torch-CPU)."""
import numpy as np
import torch

from oracle import nets, prng
from oracle import postproc as opp


def make_frames(t, h, w):
    fr = torch.zeros((t, 3, h, w), dtype=torch.float32)
    fr[:, 0, 0, 0] = torch.arange(t, dtype=torch.float32) / 1024.0        # frame id rides in one pixel
    fr[:, 1] = 0.25
    return fr


def ball_track(t, h, w):
    """(cx, cy, visible) per frame: a parabola with two invisible gaps."""
    out = []
    for f in range(t):
        cx = int(4 + (w - 10) * f / max(t - 1, 1))
        cy = int(h * 0.75 - (h * 0.5) * np.sin(np.pi * f / max(t - 1, 1)))
        vis = not (7 <= f <= 9 or 21 <= f <= 22 or f == 0)
        out.append((cx, cy, vis))
    return out


class StubTrackNet(torch.nn.Module):
    """Input (B, 3(L+1), H, W) in 'concat' layout -> (B, L, H, W) heat maps: a 3x3 blob at the ball, 0.9 inside / 0.1 outside,
    plus a smaller distractor blob on every 5th frame."""

    def __init__(self, seq_len, track):
        super().__init__()
        self.seq_len, self.track = seq_len, track

    def forward(self, x):
        b, _, h, w = x.shape
        y = torch.full((b, self.seq_len, h, w), 0.1, dtype=torch.float32, device=x.device)
        ids = (x[:, 3::3, 0, 0] * 1024.0).round().long().cpu()             # (B, L) frame ids
        for n in range(b):
            for f in range(self.seq_len):
                cx, cy, vis = self.track[int(ids[n, f])]
                if vis:
                    y[n, f, max(cy - 1, 0):cy + 2, max(cx - 1, 0):cx + 2] = 0.9
                if int(ids[n, f]) % 5 == 3:
                    y[n, f, 1:3, 1:2] = 0.8
        return y


def oracle_flow(frames, stub, sd_inpaint, seq_len, inp_len, eval_mode, batch, img_shape):
    """predict.py:120-301 restated with the oracle's functions (sliding step 1 modes only)."""
    t, _, h, w = frames.shape
    w_src, h_src = img_shape
    scaler = (w_src / opp.WIDTH, h_src / opp.HEIGHT)
    median = torch.from_numpy(np.median(frames.numpy(), 0))
    starts = list(range(0, t - seq_len + 1))
    pred = {"Frame": [], "X": [], "Y": [], "Visibility": []}
    outs = []
    for s in range(0, len(starts), batch):
        wi = starts[s:s + batch]
        x = torch.stack([torch.cat([median] + [frames[k + f] for f in range(seq_len)], 0) for k in wi], 0)
        outs.append(stub(x).numpy())
    fid = 0
    near = []                                  # frames whose ensembled heat map has a pixel within 1e-4 of the 0.5 threshold
    for ens in opp.ensemble_stream(outs, seq_len, eval_mode, len(starts)):
        n = ens.shape[0]
        ids = np.zeros((n, 1, 2), dtype=np.int64)
        ids[:, 0, 1] = np.arange(fid, fid + n)
        near += [fid + i for i in range(n) if (np.abs(ens[i] - 0.5) < 1e-4).any()]
        fid += n
        tmp = opp.predict(ids, y_pred=ens[:, None], img_scaler=scaler)
        for k in pred:
            pred[k].extend(tmp[k])
    track_pred = {k: list(v) for k, v in pred.items()}
    oracle_flow.near_threshold_frames = near
    mask = opp.generate_inpaint_mask(pred, th_h=h_src * 0.05)
    n_pts = len(pred["Frame"])
    # dataset.py:360-394,470-471: int lists concatenated onto an empty float32 array give FLOAT64; the division by the image size
    # is float64 and `coor_pred.float()` (predict.py:249) rounds once
    coor = np.stack([(np.array(pred["X"], np.float64) / w_src).astype(np.float32),
                     (np.array(pred["Y"], np.float64) / h_src).astype(np.float32)], 1)
    m = np.array(mask, np.float32).reshape(-1, 1)
    starts = list(range(0, n_pts - inp_len + 1))
    outs = []
    for s in range(0, len(starts), batch):
        wi = starts[s:s + batch]
        c = np.stack([coor[k:k + inp_len] for k in wi], 0)
        mm = np.stack([m[k:k + inp_len] for k in wi], 0)
        with torch.no_grad():
            o = nets.inpaintnet_forward(sd_inpaint, torch.from_numpy(c), torch.from_numpy(mm)).numpy()
        outs.append(opp.inpaint_blend_threshold(o, c, mm))
    final = {"Frame": [], "X": [], "Y": [], "Visibility": []}
    fid = 0
    pre_int = []                               # the floats predict.py:51 truncates: c_x * WIDTH * w_scaler, c_y * HEIGHT * h_scaler
    for ens in opp.ensemble_stream(outs, inp_len, eval_mode, len(starts)):
        th = (ens[:, 0] < opp.COOR_TH) & (ens[:, 1] < opp.COOR_TH)
        ens = ens.copy()
        ens[th] = 0
        pre_int += [(float(e[0]) * opp.WIDTH * scaler[0], float(e[1]) * opp.HEIGHT * scaler[1]) for e in ens]
        n = ens.shape[0]
        ids = np.zeros((n, 1, 2), dtype=np.int64)
        ids[:, 0, 1] = np.arange(fid, fid + n)
        fid += n
        tmp = opp.predict(ids, c_pred=ens[:, None], img_scaler=scaler)
        for k in final:
            final[k].extend(tmp[k])
    oracle_flow.pre_int = pre_int
    return track_pred, mask, final


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] with the REAL network: a TrackNet(27, 8) state whose heat maps depend on every layer of the U-Net
# yet show decisive blobs, so that integer peak-find parity is testable without a trained checkpoint (none is available
# offline).  Channels 0..7 of the full-resolution path carry relu(frame_f.R - median.R) (centre taps, identity BN) from the
# first layer through the skip connection to the head; the rest of the network keeps its synthetic calibrated weights and
# reaches those channels through the decoder with gain `deep_gain`, i.e. the 288x512 heat map is
# sigmoid(gain * (difference + deep-path term) - offset): every conv layer's arithmetic moves it, a moving bright disc
# decides it.
def detector_state(seed=31, gain=40.0, level=0.25, deep_gain=0.02):
    shapes = nets.tracknet_state_shapes(27, 8)
    sd = nets.synth_state(shapes, seed, calibrated=True)

    def ident_bn(prefix, ch):
        sd[f"{prefix}.bn.weight"][ch] = 1.0
        sd[f"{prefix}.bn.bias"][ch] = 0.0
        sd[f"{prefix}.bn.running_mean"][ch] = 0.0
        sd[f"{prefix}.bn.running_var"][ch] = 1.0

    ch = slice(0, 8)
    w = sd["down_block_1.conv_1.conv.weight"]
    w[ch] = 0.0
    for c in range(8):
        w[c, 3 * (c + 1), 1, 1] = 1.0                       # R of frame c ('concat': the median image comes first)
        w[c, 0, 1, 1] = -1.0                                # R of the median
    ident_bn("down_block_1.conv_1", ch)
    w = sd["down_block_1.conv_2.conv.weight"]
    w[ch] = 0.0
    for c in range(8):
        w[c, c, 1, 1] = 1.0
    ident_bn("down_block_1.conv_2", ch)
    w = sd["up_block_3.conv_1.conv.weight"]                 # inputs: 128 upsampled (deep path) + 64 skip (x1)
    w[ch, :128] *= deep_gain
    w[ch, 128:] = 0.0
    for c in range(8):
        w[c, 128 + c, 1, 1] = 1.0
    ident_bn("up_block_3.conv_1", ch)
    w = sd["up_block_3.conv_2.conv.weight"]
    w[ch] *= deep_gain
    for c in range(8):
        w[c, c] = 0.0
        w[c, c, 1, 1] = 1.0
    ident_bn("up_block_3.conv_2", ch)
    w = sd["predictor.weight"]
    w *= deep_gain
    for c in range(8):
        w[c, c, 0, 0] = gain
    sd["predictor.bias"][:] = -gain * level
    return sd


def disc_video(t, h=288, w=512, seed=5, sigma=3.0):
    """(t, 3, h, w) frames in [0, 1]: static texture in [0, 0.45] plus a Gaussian-profile bright disc (+0.5 on every colour)
    on a parabola with two invisible gaps; returns (frames, [(cx, cy, visible)])."""
    bg = torch.from_numpy(prng.uniform((3, h, w), seed, 0.0, 0.45))
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    frames, track = [], []
    for f in range(t):
        cx = 12.3 + (w - 25.0) * f / max(t - 1, 1)
        cy = h * 0.75 - (h * 0.5) * np.sin(np.pi * f / max(t - 1, 1)) + 0.4
        vis = not (7 <= f <= 9 or 15 <= f <= 16 or f == 0)
        fr = bg.clone()
        if vis:
            fr += torch.from_numpy(0.5 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sigma * sigma)).astype(np.float32))[None]
        frames.append(fr)
        track.append((cx, cy, vis))
    return torch.stack(frames, 0), track


def compare_final_stage(got, got_pre, want_final, want_pre, img_shape, n_masked_frames):
    """The InpaintNet stage of predict.py (213-301) against the oracle flow, given identical TrackNet-stage integers:
      * the float64 values that predict.py:51 truncates agree to 1e-6 x (source size) on EVERY coordinate -- and are EQUAL on every
        frame that no InpaintNet output reaches (blend with mask 0 returns the input exactly, and the ensemble runs in the reference's
        summation order), so those frames' integers are equal by construction;
      * the integers are compared on all 2 x T coordinates; mismatches are counted (a mismatch needs an InpaintNet-dependent float
        within 1e-6 x size of an integer).  Returns the statistics; the caller asserts the mismatch count."""
    t = len(want_final["Frame"])
    assert got["Frame"] == want_final["Frame"] and len(got_pre) == len(want_pre) == t
    worst, exact_floats, mism = 0.0, 0, []
    for f in range(t):
        for k, j in (("X", 0), ("Y", 1)):
            d = abs(got_pre[f][j] - want_pre[f][j])
            worst = max(worst, d / img_shape[j])
            exact_floats += d == 0.0
            assert d <= 1e-6 * img_shape[j], (f, k, got_pre[f][j], want_pre[f][j])
            if got[k][f] != want_final[k][f]:
                mism.append((f, k, got[k][f], want_final[k][f], want_pre[f][j]))
    return {"coordinates": 2 * t, "integer_mismatches": len(mism), "mismatch_list": mism, "pre_int_bit_equal": int(exact_floats),
            "pre_int_worst_rel": worst, "masked_frames": int(n_masked_frames)}


def check_real_network_pipeline(device, t=24, h=288, w=512, batch=6, eval_mode="weight", report=None):
    """predict_video(HIP TrackNet(27,8) + HIP InpaintNet) against oracle_flow with the ORACLE TrackNet / InpaintNet on the
    same frames.  TrackNet stage: integer dict bit-exact except frames whose ensembled oracle heat map has a pixel within
    1e-4 of the threshold (counted and reported); final stage: `compare_final_stage` -- pre-int() floats within 1e-6 x size on
    every coordinate, integer mismatches counted over all 2 x T coordinates and asserted to be ZERO."""
    from tracknetv3_amd.model import InpaintNet, TrackNet
    from tracknetv3_amd.pipeline import predict_video
    seq_len, inp_len, img_shape = 8, 16, (1920, 1080)
    frames, track = disc_video(t, h, w)
    sd_t = detector_state()
    sd_i = nets.synth_state(nets.inpaintnet_state_shapes(), 77)
    tn = TrackNet(27, 8)
    tn.load_state_dict(sd_t, strict=True)
    tn = tn.to(device).eval()
    net = InpaintNet()
    net.load_state_dict(sd_i, strict=True)
    net = net.to(device).eval()

    def oracle_tracknet(x):
        with torch.no_grad():
            return nets.tracknet_forward(sd_t, x, training=False)

    want_track, want_mask, want_final = oracle_flow(frames, oracle_tracknet, sd_i, seq_len, inp_len, eval_mode, batch, img_shape)
    near, pre_int = list(oracle_flow.near_threshold_frames), list(oracle_flow.pre_int)
    got_track = predict_video(frames.to(device), tn, None, seq_len, inp_len, "concat", eval_mode, batch, img_shape)
    n_vis = sum(want_track["Visibility"])
    assert n_vis >= t - 8, (n_vis, want_track["Visibility"])                      # the detector really detects
    for f, (cx, cy, vis) in enumerate(track):                                    # ... the disc, at the right place
        if vis and f not in near:
            assert abs(want_track["X"][f] - cx * 3.75) <= 6 and abs(want_track["Y"][f] - cy * 3.75) <= 6, (f, want_track["X"][f], cx)
    assert got_track["Frame"] == want_track["Frame"] == list(range(t))
    bad = [f for f in range(t) if f not in near and any(got_track[k][f] != want_track[k][f] for k in ("X", "Y", "Visibility"))]
    assert not bad, (bad, [(got_track["X"][f], want_track["X"][f]) for f in bad])
    assert len(near) <= t // 4, near
    dbg = {}
    got = predict_video(frames.to(device), tn, net, seq_len, inp_len, "concat", eval_mode, batch, img_shape, debug=dbg)
    stats = None
    if not near:                                   # the InpaintNet stage consumes the TrackNet-stage integers: identical inputs
        assert got["Inpaint_Mask"] == want_mask and sum(want_mask) > 0
        stats = compare_final_stage(got, dbg["pre_int"], want_final, pre_int, img_shape, sum(want_mask))
        assert stats["integer_mismatches"] == 0, stats["mismatch_list"]
        assert got["Visibility"] == want_final["Visibility"]
        assert stats["pre_int_bit_equal"] >= stats["coordinates"] // 2        # most frames never see an InpaintNet output
    if report is not None:
        report.update(near_threshold_frames=near, visible=n_vis, masked=sum(want_mask), final_stage=stats)
    return got_track, got


def check_pipeline(device, h, w, t, batch, eval_mode):
    from tracknetv3_amd.model import InpaintNet
    from tracknetv3_amd.pipeline import predict_video
    seq_len, inp_len, img_shape = 8, 16, (1920, 1080)
    frames = make_frames(t, h, w)
    track = ball_track(t, h, w)
    stub = StubTrackNet(seq_len, track)
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 77)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net = net.to(device)
    want_track, want_mask, want_final = oracle_flow(frames, stub, sd, seq_len, inp_len, eval_mode, batch, img_shape)
    got_track = predict_video(frames.to(device), stub, None, seq_len, inp_len, "concat", eval_mode, batch, img_shape)
    assert got_track == want_track                                   # integer peak-find + scaling: bit-exact
    assert len(got_track["Frame"]) == t and got_track["Frame"] == list(range(t))
    want_pre = list(oracle_flow.pre_int)
    dbg = {}
    got = predict_video(frames.to(device), stub, net, seq_len, inp_len, "concat", eval_mode, batch, img_shape, debug=dbg)
    assert got["Inpaint_Mask"] == want_mask and sum(want_mask) > 0
    # pre-int() floats within 1e-6 x size everywhere, bit-equal where no InpaintNet output reaches; ALL integers equal
    stats = compare_final_stage(got, dbg["pre_int"], want_final, want_pre, img_shape, sum(want_mask))
    assert stats["integer_mismatches"] == 0, stats["mismatch_list"]
    assert got["Visibility"] == want_final["Visibility"]
    # the non-overlap mode covers every frame exactly once as well
    no = predict_video(frames.to(device), stub, net, seq_len, inp_len, "concat", "nonoverlap", batch, img_shape)
    assert no["Frame"] == list(range(t))


def evaluate_inputs():
    """Deterministic evaluate() inputs, shared with the tests (rebuilt there from the portable PRNG)."""
    n, L = 5, 4
    idx = np.zeros((n, L, 2), dtype=np.int64)
    idx[:, :, 0] = np.arange(n)[:, None] % 2
    idx[:, :, 1] = np.arange(n * L).reshape(n, L)
    idx[4, 2:, 1] = idx[4, 1, 1]                                   # padded tail: stops after the first repeat
    y_true = np.zeros((n, L, 288, 512), dtype=np.float32)
    y_pred = (prng.uniform((n, L, 288, 512), 4100) * 0.4).astype(np.float32)
    cen = (prng.uniform((n, L, 2), 4101) * np.array([480, 260]) + 12).astype(int)
    for i in range(n):
        for f in range(L):
            cx, cy = cen[i, f]
            kind = (i * L + f) % 6
            if kind != 1 and kind != 4:                                # GT present (disc r=2.5)
                yy, xx = np.ogrid[:288, :512]
                y_true[i, f][(yy - cy) ** 2 + (xx - cx) ** 2 <= 6.25] = 1.0
            if kind in (0, 1):                                         # prediction near the centre -> TP / FP2
                y_pred[i, f, cy - 2:cy + 3, cx - 1:cx + 4] = 0.6 + 0.05 * f
            elif kind == 2:                                            # prediction far away -> FP1
                y_pred[i, f, (cy + 40) % 280:(cy + 40) % 280 + 3, (cx + 60) % 500:(cx + 60) % 500 + 3] = 0.9
            elif kind == 3:                                            # exactly `tolerance` = 4 px away (dist > tol is strict)
                y_pred[i, f, cy - 2:cy + 3, cx + 2:cx + 7] = 0.7
            # kind 4: neither (TN), kind 5: GT only (FN)
    c_true = prng.uniform((n, L, 2), 4102).astype(np.float32)
    c_pred = (c_true + (prng.uniform((n, L, 2), 4103).astype(np.float32) - 0.5) * 0.02).astype(np.float32)
    c_true[0, 1] = 0; c_pred[0, 1] = 0                                 # TN
    c_true[1, 0] = 0                                                   # FP2
    c_pred[1, 2] = 0                                                   # FN
    c_pred[2, 3] = c_true[2, 3] + 0.3                                  # FP1
    return idx, y_true, y_pred, c_true, c_pred


EVAL_CASES = {   # name -> evaluate() keyword arguments; expected dicts live in tests/golden/evaluate.npz
    "h_plain": dict(tolerance=4.), "h_full": dict(tolerance=4., img_scaler=(3.75, 3.75), output_bbox=True, output_gt=True),
    "h_tol1": dict(tolerance=1.), "c_plain": dict(tolerance=4.), "c_gt": dict(tolerance=4., img_scaler=(3.75, 3.75), output_gt=True),
}


def check_evaluate_against_golden(evaluate_fn, golden, to_dev=lambda a: a):
    """Run evaluate_fn on the shared inputs for every case and compare each list of the result with the golden."""
    idx, y_true, y_pred, c_true, c_pred = evaluate_inputs()
    for name, kw in EVAL_CASES.items():
        if name.startswith("h_"):
            got = evaluate_fn(torch.from_numpy(idx), y_true=to_dev(torch.from_numpy(y_true)), y_pred=to_dev(torch.from_numpy(y_pred)), **kw)
        else:
            got = evaluate_fn(torch.from_numpy(idx), c_true=torch.from_numpy(c_true.copy()), c_pred=torch.from_numpy(c_pred.copy()), **kw)
        want_keys = [k[len(name) + 1:] for k in golden.files if k.startswith(name + "_")]
        assert sorted(got.keys()) == sorted(want_keys), (name, sorted(got.keys()), sorted(want_keys))
        for k in want_keys:
            want = golden[f"{name}_{k}"]
            if k == "Confidence":
                assert np.array_equal(np.array(got[k], dtype=np.float32), want.astype(np.float32)), (name, k)
            else:
                assert np.array_equal(np.array(got[k]), want), (name, k, got[k], want.tolist())
