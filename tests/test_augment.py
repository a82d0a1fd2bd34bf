"""On-GPU augmentation (reference src/models/smp/dataset.py:160-207).
CPU: the host sampler draws the reference's eight transforms with its probabilities and ranges.
GPU: the kernel applies what the sampler describes (exact where the transform is exact, statistical otherwise)."""
import numpy as np
import pytest
import torch

from oct_segmentation_amd import augment as A


def test_sampler_follows_reference_probabilities_and_ranges():
    rng = np.random.default_rng(123)
    n = 20000
    _, logs = A.sample_params(n, 64, rng, return_log=True)
    freq = {k: sum(k in l for l in logs) / n for k in ('hflip', 'ssr', 'crop', 'noise', 'perspective', 'bc', 'hsv')}
    want = {'hflip': 0.50, 'ssr': 0.20, 'crop': 0.20, 'noise': 0.15, 'perspective': 0.20, 'bc': 0.15, 'hsv': 0.15}   # dataset.py:166-205
    for k in want:
        assert abs(freq[k] - want[k]) < 4 * np.sqrt(want[k] * (1 - want[k]) / n) + 1e-3, (k, freq[k])
    ssr = np.array([l['ssr'] for l in logs if 'ssr' in l])
    assert np.abs(ssr[:, 0]).max() <= 15 and 0.9 <= ssr[:, 1].min() and ssr[:, 1].max() <= 1.1 and np.abs(ssr[:, 2:]).max() <= 0.0625 * 64
    crop = np.array([l['crop'] for l in logs if 'crop' in l])
    assert crop[:, 2].min() >= int(0.8 * 64) and crop[:, 2].max() <= int(0.9 * 64) and crop[:, 3].min() >= int(0.8 * 64)
    assert (crop[:, 0] + crop[:, 2] <= 64).all() and (crop[:, 1] + crop[:, 3] <= 64).all()
    noise = np.array([l['noise'] for l in logs if 'noise' in l]) ** 2
    assert 1.5 <= noise.min() and noise.max() <= 6.5
    bc = np.array([l['bc'] for l in logs if 'bc' in l])
    assert np.abs(bc[:, 0] - 1).max() <= 0.15 and np.abs(bc[:, 1]).max() <= 0.15
    hsv = np.array([l['hsv'] for l in logs if 'hsv' in l])
    assert np.abs(hsv[:, 0]).max() <= 15 and np.abs(hsv[:, 1]).max() <= 20 and np.abs(hsv[:, 2]).max() <= 15


def test_homography_solver_and_packing():
    src = np.array([[0, 0], [63, 0], [63, 63], [0, 63]], dtype=np.float64)
    dst = src + np.array([[3, 2], [-4, 1], [-2, -5], [1, -3]])
    H = A._homography_from_points(src, dst)
    for (x, y), (u, v) in zip(src, dst):
        w = H @ np.array([x, y, 1.0])
        assert np.allclose(w[:2] / w[2], [u, v], atol=1e-9)
    row = A.pack_params(H, alpha=1.1, beta=-0.05, sigma=2.0, seed=77, hue=3, sat=-4, val=5, hsv_on=True)
    Hinv = row[0:9].reshape(3, 3).astype(np.float64)
    assert np.allclose(Hinv @ H / (Hinv @ H)[2, 2], np.eye(3), atol=1e-4)
    assert row[9] == np.float32(1.1) and row[16] == 1.0 and row[12:13].view(np.uint32)[0] == 77


# ------------------------------------------------------------------------------------------------ GPU
def _frames(B=3, C=2, S=48, seed=0):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (B, 3, S, S), generator=g).float()
    mask = (torch.rand(B, C, S, S, generator=g) > 0.6).float()
    return img, mask


@pytest.mark.gpu
def test_identity_flip_shift_and_crop_are_exact(cuda):
    S = 48
    img, mask = _frames(4, 2, S)
    flip = np.array([[-1, 0, S - 1], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
    shift = A._translate(5, -3)
    rect = (4, 6, 4 + 40, 6 + 38)                                   # crop 38 x 40 at (y0=2, x0=7), padded to (6, 4)
    crop = A._translate(4 - 7, 6 - 2)
    p = np.stack([A.pack_params(np.eye(3)), A.pack_params(flip), A.pack_params(shift), A.pack_params(crop, rect=rect)])
    out, mout = A.augment(img.to(cuda), mask.to(cuda), p)
    out, mout = out.cpu(), mout.cpu()
    assert torch.equal(out[0], img[0]) and torch.equal(mout[0], mask[0])
    assert torch.equal(out[1], img[1].flip(-1)) and torch.equal(mout[1], mask[1].flip(-1))
    want = torch.zeros_like(img[2]); want[:, :S - 3, 5:] = img[2][:, 3:, :S - 5]      # out(x, y) = in(x - 5, y + 3)
    assert torch.equal(out[2], want)
    want = torch.zeros_like(img[3]); want[:, 6:44, 4:44] = img[3][:, 2:40, 7:47]       # RandomCrop + centred PadIfNeeded
    wm = torch.zeros_like(mask[3]); wm[:, 6:44, 4:44] = mask[3][:, 2:40, 7:47]
    assert torch.equal(out[3], want) and torch.equal(mout[3], wm)


@pytest.mark.gpu
def test_warp_matches_a_torch_bilinear_gather_and_masks_stay_binary(cuda):
    S = 64
    img, mask = _frames(6, 3, S, seed=2)
    rng = np.random.default_rng(5)
    rows, Ms = [], []
    for _ in range(6):
        angle, scale = rng.uniform(-15, 15), 1 + rng.uniform(-0.1, 0.1)
        a, c = np.deg2rad(angle), S / 2 - 0.5
        R = np.array([[scale * np.cos(a), scale * np.sin(a), 0], [-scale * np.sin(a), scale * np.cos(a), 0], [0, 0, 1]])
        M = A._translate(rng.uniform(-4, 4), rng.uniform(-4, 4)) @ A._translate(c, c) @ R @ A._translate(-c, -c)
        jit = np.abs(rng.normal(0, 0.07, (4, 2))) * S
        src = np.array([[0, 0], [S - 1, 0], [S - 1, S - 1], [0, S - 1]], dtype=np.float64)
        M = A._homography_from_points(src + jit * np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]]), src) @ M
        Ms.append(M); rows.append(A.pack_params(M))
    out, mout = A.augment(img.to(cuda), mask.to(cuda), np.stack(rows))
    out, mout = out.cpu(), mout.cpu()
    assert set(np.unique(mout.numpy())) <= {0.0, 1.0}
    ys, xs = torch.meshgrid(torch.arange(S, dtype=torch.float64), torch.arange(S, dtype=torch.float64), indexing='ij')
    for n, M in enumerate(Ms):
        Hi = torch.tensor(np.linalg.inv(M))
        w = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
        sx, sy = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / w, (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / w
        grid = torch.stack([sx / (S - 1) * 2 - 1, sy / (S - 1) * 2 - 1], dim=-1)[None].float()
        ref = torch.nn.functional.grid_sample(img[n:n + 1], grid, mode='bilinear', padding_mode='zeros', align_corners=True)[0]
        # the kernel rounds to the uint8 grid: half a grey level, plus float slack on the coordinates
        assert (out[n] - ref).abs().max().item() <= 0.5 + 0.3


def test_oracle_hsv_8bit_known_answers():
    """The numpy restatement of OpenCV's 8-bit RGB<->HSV (oracle/augment_ref.py) on colours whose HSV is documented: H in half
    degrees [0, 180), S and V in [0, 255]; greys have S = 0 and round-trip exactly; saturated primaries round-trip exactly."""
    from oracle.augment_ref import hsv2rgb_u8, rgb2hsv_u8, shift_hsv_uint8
    rgb = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255], [128, 128, 128], [0, 0, 0], [255, 255, 255],
                    [200, 100, 50]], dtype=np.uint8)
    hsv = rgb2hsv_u8(rgb)
    want = np.array([[0, 255, 255], [60, 255, 255], [120, 255, 255], [30, 255, 255], [90, 255, 255], [150, 255, 255], [0, 0, 128], [0, 0, 0], [0, 0, 255],
                     [10, 191, 200]], dtype=np.uint8)
    assert np.array_equal(hsv, want)
    assert np.array_equal(hsv2rgb_u8(hsv[:9]), rgb[:9])
    assert np.abs(hsv2rgb_u8(hsv[9:]).astype(int) - rgb[9:].astype(int)).max() <= 2      # H is quantised to 2 degrees
    # a hue shift of +60 (half-degree units) turns red into blue... in the frame albumentations believes to be RGB
    assert np.array_equal(shift_hsv_uint8(rgb[:1], 60, 0, 0), rgb[1:2])
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    assert np.array_equal(shift_hsv_uint8(x, 0, 0, 0), hsv2rgb_u8(rgb2hsv_u8(x)))
    assert np.abs(hsv2rgb_u8(rgb2hsv_u8(x)).astype(int) - x.astype(int)).max() <= 6        # the 8-bit round trip is lossy by design


@pytest.mark.gpu
def test_photometric_ops_match_the_uint8_chain(cuda):
    """GaussNoise / RandomBrightnessContrast / HueSaturationValue of the kernel against the oracle's restatement of the reference's
    uint8 stages (albumentations 1.4.3 over OpenCV 8-bit HSV; oracle/augment_ref.py): brightness-contrast and every hue / saturation /
    value shift within one grey level (float vs the fixed-point reciprocal tables), noise by its statistics."""
    from oracle.augment_ref import brightness_contrast_uint8, shift_hsv_uint8
    S = 64
    img, mask = _frames(8, 1, S, seed=3)
    rng = np.random.default_rng(4)
    shifts = [(0.0, 0.0, 0.0), (10.0, -15.0, 12.0), (-14.3, 19.6, -7.2), (3.7, 0.0, 0.0), (0.0, -20.0, 0.0), (0.0, 0.0, 15.0)]
    rows = [A.pack_params(np.eye(3), alpha=1.12, beta=-0.08), A.pack_params(np.eye(3), sigma=2.0, seed=4242)]
    rows += [A.pack_params(np.eye(3), hue=h, sat=s_, val=v, hsv_on=True) for h, s_, v in shifts]
    out, _ = A.augment(img.to(cuda), mask.to(cuda), np.stack(rows))
    out = out.cpu()
    hwc = lambda t: t.permute(1, 2, 0).numpy().astype(np.uint8)     # noqa: E731  (channel 0 = B of the reference's BGR frame)
    want = brightness_contrast_uint8(hwc(img[0]), 1.12, -0.08)
    assert np.abs(hwc(out[0]).astype(int) - want.astype(int)).max() <= 1
    assert (hwc(out[0]) != want).mean() < 0.01
    d = (out[1] - img[1]).flatten()                                 # GaussNoise: N(0, sigma^2) added, clipped, TRUNCATED (mean -0.5)
    inner = (img[1].flatten() > 8) & (img[1].flatten() < 247)
    assert abs(d[inner].mean().item() + 0.5) < 0.06 and abs(d[inner].std().item() - (4.0 + 1.0 / 12) ** 0.5) < 0.1
    for k, (h, s_, v) in enumerate(shifts):
        got = hwc(out[2 + k])
        ref = shift_hsv_uint8(hwc(img[2 + k]), h, s_, v)
        diff = np.abs(got.astype(int) - ref.astype(int))
        assert diff.max() <= 1, (h, s_, v, diff.max())
        assert (diff > 0).mean() < 0.02, (h, s_, v, (diff > 0).mean())
    assert out.min().item() >= 0 and out.max().item() <= 255 and torch.equal(out, out.round())
    _ = rng


@pytest.mark.gpu
def test_distribution_against_the_sequential_uint8_chain(cuda):
    """The kernel's single resampling against the oracle's restatement of the reference's SEQUENTIAL chain (every stage its own uint8
    image): the same per-frame decisions (sample_frame), 400 OCT-shaped frames.  Frames without a resampled stage must agree
    to one grey level (flip, crop + pad, brightness-contrast, HSV); over all frames the per-channel mean and standard deviation
    agree within 1 % and the masks within 1 % of their area."""
    from oracle.augment_ref import apply_chain
    from synth import make_batch
    S, N = 64, 400
    rng = np.random.default_rng(2024)
    img, mask = make_batch(N, 2, S, seed=77)
    rows, logs = [], []
    for _ in range(N):
        M, post, rect, alpha, beta, sigma, seed, hue, sat, val, hsv_on, log = A.sample_frame(S, rng)
        rows.append(A.pack_params(M, alpha, beta, sigma, seed, hue, sat, val, hsv_on, post, rect)); logs.append(log)
    out, mout = A.augment(img.to(cuda), mask.to(cuda), np.stack(rows))
    out, mout = out.cpu().numpy(), mout.cpu().numpy()
    noise_rng = np.random.default_rng(5)
    ref_i, ref_m = np.empty_like(out), np.empty_like(mout)
    exact = 0
    for n in range(N):
        i8 = img[n].permute(1, 2, 0).numpy().astype(np.uint8)
        m8 = mask[n].permute(1, 2, 0).numpy().astype(np.uint8)
        g = noise_rng.normal(0.0, logs[n]['noise'], size=(S, S, 3)) if 'noise' in logs[n] else None
        ri, rm = apply_chain(i8, m8, logs[n], S, g)
        ref_i[n], ref_m[n] = ri.transpose(2, 0, 1), rm.transpose(2, 0, 1)
        if not ({'ssr', 'perspective', 'noise'} & set(logs[n])):
            assert np.abs(out[n] - ref_i[n]).max() <= 1.0, logs[n]
            assert np.array_equal(mout[n], ref_m[n].astype(np.float32)), logs[n]
            exact += 1
    assert exact > N // 3
    for c in range(3):
        a, b = out[:, c].astype(np.float64), ref_i[:, c].astype(np.float64)
        assert abs(a.mean() - b.mean()) <= 0.01 * b.mean() + 0.05, (c, a.mean(), b.mean())
        assert abs(a.std() - b.std()) <= 0.01 * b.std() + 0.05, (c, a.std(), b.std())
    assert abs(mout.mean() - ref_m.mean()) <= 0.01 * ref_m.mean()
    # per frame, the resampled stages differ by interpolation only
    per_frame = np.abs(out - ref_i).mean(axis=(1, 2, 3))
    noisy = np.array(['noise' in l for l in logs])
    assert per_frame[~noisy].max() < 3.0 and np.median(per_frame[~noisy]) < 0.5


@pytest.mark.gpu
def test_sampled_batch_runs_and_keeps_the_contract(cuda):
    S = 96
    img, mask = _frames(16, 4, S, seed=9)
    p = A.sample_params(16, S, np.random.default_rng(11))
    out, mout = A.augment(img.to(cuda), mask.to(cuda), p)
    assert out.shape == img.shape and mout.shape == mask.shape
    assert out.min().item() >= 0 and out.max().item() <= 255 and torch.equal(out, out.round())
    assert set(np.unique(mout.cpu().numpy())) <= {0.0, 1.0}
    with pytest.raises(ValueError):
        A.augment(img.to(cuda), mask.to(cuda), p[:3])
