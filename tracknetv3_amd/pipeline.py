"""End-to-end trajectory prediction on a frame stream: the `__main__` of the reference's predict.py (predict.py:120-301)
with every arithmetic stage on the device -- window assembly, TrackNet, temporal ensemble, threshold + peak-find,
inpaint-mask scan, InpaintNet, blend + COOR_TH threshold, coordinate ensemble, final coordinates.

`frames` is either the source-resolution stream as uint8 (T, H, W, 3) RGB -- then the median background, the
Pillow-exact bicubic resize to 288x512 and the `/255` normalisation run on the device too (`preprocess.py`) -- or
already-resized float frames (T, 3, 288, 512) in [0, 1].  Video decoding stays outside.
"""
import torch

from . import postprocess as pp
from .model import no_infer_split as _no_infer_split
from .utils.general import HEIGHT, WIDTH


def _windows(num_frames, seq_len, step, padding):
    """Frame indices of every input window, as dataset.py:329-354 builds them (padding repeats the last frame)."""
    out = []
    last = -1
    for i in range(0, num_frames, step):
        idx = []
        for f in range(seq_len):
            if i + f < num_frames:
                idx.append(i + f)
                last = i + f
            elif padding:
                idx.append(last)
            else:
                break
        if len(idx) == seq_len:
            out.append(idx)
    return torch.tensor(out, dtype=torch.long).reshape(-1, seq_len)


def _assemble(frames, median, widx, bg_mode):
    """(B, L) frame indices -> network input (B, C, H, W): 'concat' puts the median image first (dataset.py:455-456)."""
    b, l = widx.shape
    x = frames[widx.to(frames.device)].reshape(b, l * frames.shape[1], frames.shape[-2], frames.shape[-1])
    if bg_mode == "concat":
        x = torch.cat((median.unsqueeze(0).expand(b, -1, -1, -1), x), dim=1)
    return x.contiguous()


def _index_tensor(widx):
    i = torch.zeros(widx.shape + (2,), dtype=torch.long)
    i[..., 1] = widx
    return i


def _np_median0(frames):
    """np.median(frames, 0) for float frames: the middle value, or the mean of the two middle values when the count is even."""
    t = int(frames.shape[0])
    if t % 2:
        return frames.median(dim=0).values
    lo = torch.kthvalue(frames, t // 2, dim=0).values
    hi = torch.kthvalue(frames, t // 2 + 1, dim=0).values
    return (lo + hi) * 0.5


_SIDE_STREAMS = {}
# uint8 source streams: resize lazily on the batches' streams (preprocess.LazyResizer) instead of everything up front.  TNV3_LAZY_RESIZE=0: up front.
LAZY_RESIZE = __import__("os").environ.get("TNV3_LAZY_RESIZE", "1") != "0"


def _side_streams(dev, n):
    """Side streams are kept per device: the caching allocator pools memory per stream, so fresh streams on every call
    would pay a hipMalloc for each activation of the first forwards."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    have = _SIDE_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(dev))
    return have[:n]


def _tracknet_batches(tracknet, frames, median, widx, bg_mode, batch_size, n_streams=2, ensure=None):
    """Yields (window-index batch, heat maps) in order, keeping up to `n_streams` TrackNet forwards in flight on side
    HIP streams: windows are independent, so the next batch's launches fill the CUs that the 45/48 tail of every
    batch-sized conv launch leaves idle (+6 % measured), and they run while the host post-processes the previous batch.
    The consumer's stream waits on each batch's completion event before it reads the heat maps."""
    dev = frames.device
    starts = list(range(0, int(widx.shape[0]), batch_size))
    # ensure(lo, hi): makes frames lo .. hi - 1 of `frames` valid on the current stream (preprocess.LazyResizer: the resize of a uint8 source
    # stream, chunk by chunk, on the stream of the batch that needs it first)
    if dev.type != "cuda" or n_streams < 2 or len(starts) < 2:
        for s in starts:
            wi = widx[s:s + batch_size]
            if ensure is not None:
                ensure(int(wi.min()), int(wi.max()) + 1)
            yield wi, tracknet(_assemble(frames, median, wi, bg_mode))
        return
    main = torch.cuda.current_stream(dev)
    if hasattr(tracknet, "prepare_eval"):
        tracknet.prepare_eval()                          # cached operands are built here, before the side streams read them
    side = _side_streams(dev, n_streams)
    ready = torch.cuda.Event()
    ready.record(main)                                   # frames / median were produced on the consumer's stream
    pending = []

    def launch(k):
        wi = widx[starts[k]:starts[k] + batch_size]
        st = side[k % n_streams]
        with torch.cuda.stream(st), _no_infer_split():     # batches already overlap here: no second split inside each
            st.wait_event(ready)
            if ensure is not None:
                ensure(int(wi.min()), int(wi.max()) + 1)
            y = tracknet(_assemble(frames, median, wi, bg_mode))
            done = torch.cuda.Event()
            done.record(st)
        pending.append((wi, y, done))

    for k in range(min(n_streams, len(starts))):
        launch(k)
    nxt = len(pending)
    while pending:
        wi, y, done = pending.pop(0)
        main.wait_event(done)
        y.record_stream(main)                            # allocated on a side stream, consumed (and freed) on this one
        yield wi, y
        if nxt < len(starts):
            launch(nxt)
            nxt += 1


@torch.no_grad()
def predict_video(frames, tracknet, inpaintnet=None, tracknet_seq_len=8, inpaintnet_seq_len=16, bg_mode="concat",
                  eval_mode="weight", batch_size=16, img_shape=None, median=None, debug=None):
    """Returns the reference's pred_dict {'Frame','X','Y','Visibility'} (+ 'Inpaint_Mask' when InpaintNet runs)
    for a (T, 3, 288, 512) frame tensor.  img_shape = (w, h) of the source video (default: the network resolution).
    debug: an optional dict that receives 'pre_int' -- per output frame the two float64 values that predict.py:51 truncates
    with int() in the InpaintNet stage (the parity tests compare them with the oracle's before looking at the integers)."""
    if eval_mode not in ("nonoverlap", "average", "weight"):
        raise ValueError("Invalid mode")
    ensure = None
    if frames.dtype == torch.uint8:                      # source-resolution (T, H, W, 3) stream: preprocess on the device
        from . import preprocess
        if img_shape is None:
            img_shape = (int(frames.shape[2]), int(frames.shape[1]))
        if bg_mode in ("", None, "concat") and frames.is_cuda and LAZY_RESIZE:
            # the median up front; the bicubic resize per chunk of frames on the stream of the TrackNet batch that needs it first
            lazy = preprocess.LazyResizer(frames, bg_mode)
            frames, med, ensure = lazy.out, lazy.median, lazy.ensure
        else:
            frames, med = preprocess.preprocess_video(frames, bg_mode)
        if median is None:
            median = med
    t = int(frames.shape[0])
    w_src, h_src = img_shape if img_shape is not None else (WIDTH, HEIGHT)
    img_scaler = (w_src / WIDTH, h_src / HEIGHT)
    if bg_mode == "concat" and median is None:
        # np.median's semantics (dataset.py:105: the MEAN of the two middle values for an even frame count), not torch's lower median
        median = _np_median0(frames)
    tracknet.eval()
    pred = {"Frame": [], "X": [], "Y": [], "Visibility": []}
    seq_len = tracknet_seq_len

    if eval_mode == "nonoverlap":
        widx = _windows(t, seq_len, seq_len, padding=True)
        for wi, y in _tracknet_batches(tracknet, frames, median, widx, bg_mode, batch_size, ensure=ensure):
            tmp = pp.predict(_index_tensor(wi), y_pred=y, img_scaler=img_scaler)
            for k in pred:
                pred[k].extend(tmp[k])
    else:
        widx = _windows(t, seq_len, 1, padding=False)
        num_sample = int(widx.shape[0])
        stream = pp.EnsembleStream(seq_len, eval_mode, num_sample)
        frame_id = 0
        for wi, y in _tracknet_batches(tracknet, frames, median, widx, bg_mode, batch_size, ensure=ensure):
            ens = stream.push(y)                                                       # (n_frames, H, W), on device
            n = int(ens.shape[0])
            ids = torch.zeros((n, 1, 2), dtype=torch.long)
            ids[:, 0, 1] = torch.arange(frame_id, frame_id + n)
            frame_id += n
            tmp = pp.predict(ids, y_pred=ens.unsqueeze(1), img_scaler=img_scaler)
            for k in pred:
                pred[k].extend(tmp[k])

    if inpaintnet is None:
        return pred

    # ---- TrackNetV3 = TrackNet + InpaintNet (predict.py:213-301)
    inpaintnet.eval()
    dev = frames.device
    pred["Inpaint_Mask"] = pp.generate_inpaint_mask(pred, th_h=h_src * 0.05)
    seq_len = inpaintnet_seq_len
    batch_size = max(batch_size, 1024)       # trajectory windows are 48 floats each: the reference's batch only bounds launch count
    n_pts = len(pred["Frame"])
    # dataset.py:360-394,470-471: the integer lists are concatenated onto an empty float32 array -> a FLOAT64 array, divided by the
    # image size in float64, and only `coor_pred.float()` (predict.py:221,249) rounds to fp32
    coor_all = torch.tensor([pred["X"], pred["Y"]], dtype=torch.float64).t().contiguous()      # source-pixel units
    coor_all[:, 0] /= w_src
    coor_all[:, 1] /= h_src
    coor_all = coor_all.float()
    mask_all = torch.tensor(pred["Inpaint_Mask"], dtype=torch.float32).reshape(-1, 1)
    out = {"Frame": [], "X": [], "Y": [], "Visibility": []}

    if eval_mode == "nonoverlap":
        widx = _windows(n_pts, seq_len, seq_len, padding=True)
        for s in range(0, widx.shape[0], batch_size):
            wi = widx[s:s + batch_size]
            c, m = coor_all[wi].to(dev), mask_all[wi].to(dev)
            ci = pp.inpaint_blend_threshold(inpaintnet(c, m), c, m)
            tmp = pp.predict(_index_tensor(wi), c_pred=ci, img_scaler=img_scaler)
            if debug is not None:      # nonoverlap: windows tile the frame list; a padded tail repeats the last frame and is cut by predict()
                flat = ci.detach().double().cpu().reshape(-1, 2)[:len(tmp["Frame"])]
                debug.setdefault("pre_int", []).extend(
                    (float(v[0]) * WIDTH * img_scaler[0], float(v[1]) * HEIGHT * img_scaler[1]) for v in flat)
            for k in out:
                out[k].extend(tmp[k])
    else:
        widx = _windows(n_pts, seq_len, 1, padding=False)
        num_sample = int(widx.shape[0])
        stream = pp.EnsembleStream(seq_len, eval_mode, num_sample)
        frame_id = 0
        for s in range(0, num_sample, batch_size):
            wi = widx[s:s + batch_size]
            c, m = coor_all[wi].to(dev), mask_all[wi].to(dev)
            ci = pp.inpaint_blend_threshold(inpaintnet(c, m), c, m)
            ens = stream.push(ci)                                                               # (n_frames, 2)
            th = (ens[:, 0] < pp.COOR_TH) & (ens[:, 1] < pp.COOR_TH)
            ens = ens.clone()
            ens[th] = 0.0
            n = int(ens.shape[0])
            ids = torch.zeros((n, 1, 2), dtype=torch.long)
            ids[:, 0, 1] = torch.arange(frame_id, frame_id + n)
            frame_id += n
            tmp = pp.predict(ids, c_pred=ens.unsqueeze(1), img_scaler=img_scaler)
            if debug is not None:
                debug.setdefault("pre_int", []).extend(
                    (float(v[0]) * WIDTH * img_scaler[0], float(v[1]) * HEIGHT * img_scaler[1]) for v in ens.detach().double().cpu())
            for k in out:
                out[k].extend(tmp[k])
    out["Inpaint_Mask"] = pred["Inpaint_Mask"]
    return out
