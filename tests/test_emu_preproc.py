"""CPU: frame-preprocessing kernels through the emulator vs the oracle restatement, and the oracle vs LIVE Pillow."""
import numpy as np
import pytest
import torch

from oracle import preproc as opre


def test_oracle_resize_is_bit_exact_with_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(0)
    for (h, w, oh, ow) in ((270, 480, 72, 128), (108, 192, 72, 128), (72, 128, 72, 128), (50, 75, 72, 128), (1080, 1920, 288, 512)):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        img[: h // 4, : w // 3] = 255
        img[h // 2:, w // 2:] = 0
        assert np.array_equal(opre.resize_bicubic_u8(img, ow, oh), np.array(Image.fromarray(img).resize(size=(ow, oh))))
    g = rng.randint(0, 256, (100, 160)).astype(np.uint8)
    assert np.array_equal(opre.resize_bicubic_u8(g[..., None], 64, 36)[..., 0], np.array(Image.fromarray(g).resize(size=(64, 36))))


def test_product_coefficient_tables_equal_oracle():
    from tracknetv3_amd.preprocess import resample_coeffs
    for (a, b) in ((1920, 512), (1080, 288), (480, 128), (75, 128), (128, 128)):
        for x, y in zip(resample_coeffs(a, b), opre.resample_coeffs(a, b)):
            assert np.array_equal(x, y)


def test_resize_and_median_emulated_vs_oracle(emu):
    from tracknetv3_amd import preprocess as pre
    rng = np.random.RandomState(1)
    for (h, w, oh, ow) in ((54, 96, 16, 32), (20, 30, 16, 32), (16, 32, 16, 32)):
        fr = rng.randint(0, 256, (3, h, w, 3)).astype(np.uint8)
        fr[0, :5] = 255
        f32, u8 = pre.resize_frames(torch.from_numpy(fr), oh, ow, want_f32=True, want_u8=True)
        for k in range(3):
            want = opre.resize_bicubic_u8(fr[k], ow, oh)
            assert np.array_equal(u8[k].numpy(), want)
            assert np.array_equal(f32[k].numpy(), opre.normalise_u8(np.moveaxis(want, -1, 0)))
    for t, (mh, mw) in ((1, (6, 10)), (2, (6, 10)), (7, (5, 7)), (16, (6, 10)), (33, (6, 10)), (100, (4, 6)), (46, (5, 7))):
        fr = rng.randint(0, 256, (t, mh, mw, 3)).astype(np.uint8)   # 6x10x3 bytes: radix-select kernel; 5x7x3: LDS-histogram kernel
        fr[:, 0, 0] = 200                                            # constant pixel
        fr[: t // 2, 0, 1] = 0
        fr[t // 2:, 0, 1] = 255                                      # bimodal: even T averages 0 and 255 -> 127
        fr[:, 1, 0] = rng.randint(96, 112, (t, 3))                   # all values inside one high nibble
        fr[:, 1, 1] = rng.randint(110, 114, (t, 3))                  # straddling a high-nibble boundary (111 | 112)
        got = pre.median_background(torch.from_numpy(fr)).numpy()
        assert np.array_equal(got, opre.median_u8(fr)), t
        m2 = pre.median_background(torch.from_numpy(fr), doubled=True).numpy()
        assert np.array_equal(m2.astype(np.float64) / 2, np.median(fr, 0)), t


@pytest.mark.parametrize("cus", [1, 2])
def test_resize_persistent_horizontal_pass_walks_many_row_groups_emulated(emu, monkeypatch, cus):
    """Round 6's horizontal pass is a persistent workgroup (coefficients in registers, the next group of four source rows fetched while the current
    one is computed, two LDS stages): with 1 / 2 emulated CUs (3 / 6 workgroups) every workgroup walks several row groups, the last group is
    partial (rows % 4 != 0), and 40 output columns leave most threads of the second column set idle.  Bit-exact against Pillow's algorithm."""
    from tracknetv3_amd import preprocess as pre
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    rng = np.random.RandomState(7)
    for (f, h, w, oh, ow) in ((3, 54, 96, 16, 32), (2, 23, 64, 9, 40), (1, 37, 160, 10, 300), (2, 61, 64, 7, 48), (1, 9, 32, 20, 16)):      # (9 -> 20 rows: up-scaling, five taps)
        fr = rng.randint(0, 256, (f, h, w, 3)).astype(np.uint8)
        fr[0, :5] = 255
        fr[-1, -3:] = 0
        f32, u8 = pre.resize_frames(torch.from_numpy(fr), oh, ow, want_f32=True, want_u8=True)
        for k in range(f):
            want = opre.resize_bicubic_u8(fr[k], ow, oh)
            assert np.array_equal(u8[k].numpy(), want), (f, h, w, oh, ow, k)
            assert np.array_equal(f32[k].numpy(), opre.normalise_u8(np.moveaxis(want, -1, 0)))


def test_preprocess_video_matches_reference_dataset_layout(emu):
    """median first for 'concat', frame channels RGB-major, /255 -- vs the oracle's restatement of dataset.py:427-461."""
    from tracknetv3_amd import preprocess as pre
    from tracknetv3_amd.pipeline import _assemble, _windows
    rng = np.random.RandomState(2)
    fr = rng.randint(0, 256, (6, 40, 64, 3)).astype(np.uint8)
    import tracknetv3_amd.preprocess as p
    frames, med = None, None
    old = (p.HEIGHT, p.WIDTH)
    try:
        frames = torch.cat([pre.resize_frames(torch.from_numpy(fr), 16, 32)], 0)
        med = pre.resize_frames(pre.median_background(torch.from_numpy(fr)).unsqueeze(0), 16, 32)[0]
    finally:
        p.HEIGHT, p.WIDTH = old
    widx = _windows(6, 3, 1, padding=False)
    x = _assemble(frames, med, widx, "concat").numpy()
    want = opre.tracknet_input_from_frames(fr, [0, 1, 2, 3], 3, "concat", height=16, width=32)
    assert x.shape == want.shape == (4, 12, 16, 32) and np.array_equal(x, want)
    x0 = _assemble(frames, None, widx, "").numpy()
    assert np.array_equal(x0, opre.tracknet_input_from_frames(fr, [0, 1, 2, 3], 3, "", height=16, width=32))


def test_difference_frame_modes_emulated_vs_oracle(emu):
    """bg_mode 'subtract' / 'subtract_concat' (dataset.py:439-446): float median (halves), channel sum up to 765,
    uint8 truncation + wrap -- bit-exact in integer arithmetic; odd and even frame counts."""
    from tracknetv3_amd import preprocess as pre
    from tracknetv3_amd.pipeline import _assemble, _windows
    rng = np.random.RandomState(4)
    for t in (6, 7):
        fr = rng.randint(0, 256, (t, 20, 24, 3)).astype(np.uint8)
        fr[: t // 2, 2:6, 2:6] = 0
        fr[t // 2:, 2:6, 2:6] = 255                                   # big differences: sums above 255 wrap
        med64 = np.median(fr, 0)
        m2 = pre.median_background(torch.from_numpy(fr), doubled=True).numpy()
        assert np.array_equal(m2.astype(np.float64) / 2.0, med64)
        d = pre.difference_frames(torch.from_numpy(fr), torch.from_numpy(m2)).numpy()[..., 0]
        for k in range(t):
            assert np.array_equal(d[k], opre.diff_frame_u8(fr[k], med64)), (t, k)
    import tracknetv3_amd.preprocess as p
    fr = rng.randint(0, 256, (6, 40, 64, 3)).astype(np.uint8)
    for mode, c in (("subtract", 1), ("subtract_concat", 4)):
        med2 = pre.median_background(torch.from_numpy(fr), doubled=True)
        planes = []
        if mode != "subtract":
            planes.append(pre.resize_frames(torch.from_numpy(fr), 16, 32))
        planes.append(pre.resize_frames(pre.difference_frames(torch.from_numpy(fr), med2), 16, 32))
        frames = planes[0] if len(planes) == 1 else torch.cat(planes, 1)
        assert frames.shape == (6, c, 16, 32)
        x = _assemble(frames, None, _windows(6, 3, 1, False), mode).numpy()
        want = opre.tracknet_input_from_frames(fr, [0, 1, 2, 3], 3, mode, height=16, width=32)
        assert x.shape == want.shape == (4, 3 * c, 16, 32) and np.array_equal(x, want), mode
