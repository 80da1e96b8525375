"""Shared by the CPU (emulator) and GPU pipeline tests: a synthetic 'TrackNet' stub that paints clean heat maps from
frame ids carried in the frames, and the oracle-side restatement of predict.py's flow (numpy / torch-CPU)."""
import numpy as np
import torch

from oracle import nets
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
    median = frames.median(dim=0).values
    starts = list(range(0, t - seq_len + 1))
    pred = {"Frame": [], "X": [], "Y": [], "Visibility": []}
    outs = []
    for s in range(0, len(starts), batch):
        wi = starts[s:s + batch]
        x = torch.stack([torch.cat([median] + [frames[k + f] for f in range(seq_len)], 0) for k in wi], 0)
        outs.append(stub(x).numpy())
    fid = 0
    for ens in opp.ensemble_stream(outs, seq_len, eval_mode, len(starts)):
        n = ens.shape[0]
        ids = np.zeros((n, 1, 2), dtype=np.int64)
        ids[:, 0, 1] = np.arange(fid, fid + n)
        fid += n
        tmp = opp.predict(ids, y_pred=ens[:, None], img_scaler=scaler)
        for k in pred:
            pred[k].extend(tmp[k])
    track_pred = {k: list(v) for k, v in pred.items()}
    mask = opp.generate_inpaint_mask(pred, th_h=h_src * 0.05)
    n_pts = len(pred["Frame"])
    coor = np.stack([np.array(pred["X"], np.float32) / w_src, np.array(pred["Y"], np.float32) / h_src], 1)
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
    for ens in opp.ensemble_stream(outs, inp_len, eval_mode, len(starts)):
        th = (ens[:, 0] < opp.COOR_TH) & (ens[:, 1] < opp.COOR_TH)
        ens = ens.copy()
        ens[th] = 0
        n = ens.shape[0]
        ids = np.zeros((n, 1, 2), dtype=np.int64)
        ids[:, 0, 1] = np.arange(fid, fid + n)
        fid += n
        tmp = opp.predict(ids, c_pred=ens[:, None], img_scaler=scaler)
        for k in final:
            final[k].extend(tmp[k])
    return track_pred, mask, final


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
    got = predict_video(frames.to(device), stub, net, seq_len, inp_len, "concat", eval_mode, batch, img_shape)
    assert got["Inpaint_Mask"] == want_mask and sum(want_mask) > 0
    assert got["Frame"] == want_final["Frame"] and got["Visibility"] == want_final["Visibility"]
    # coordinates pass through fp32 InpaintNet arithmetic and int() truncation: allow one source pixel
    assert max(abs(a - b) for a, b in zip(got["X"], want_final["X"])) <= 1
    assert max(abs(a - b) for a, b in zip(got["Y"], want_final["Y"])) <= 1
    # the non-overlap mode covers every frame exactly once as well
    no = predict_video(frames.to(device), stub, net, seq_len, inp_len, "concat", "nonoverlap", batch, img_shape)
    assert no["Frame"] == list(range(t))
