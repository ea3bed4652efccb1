"""Synthetic sequences with analytic ground-truth motion for benchmarks and
full-size property tests.

Scene: a smooth gradient plus 24 oriented sinusoid patches and 8 soft-edged
discs, translated by (0.7, -0.4) pixels (columns, rows) per frame, with one
disc moving independently by (-1.3, 0.9) per frame.  The flow handed to the
solver is that motion in the reference's 'backward' convention
u_t(x) ~ u_{t+1}(x + v_t(x)), component 0 along columns, component 1 along
rows (helpers/constructWarpFF.m:5, variants/testfile_gt_sintel.m:36-38).  The
low-resolution data is the model's own forward operator (block mean of the
7x7 Gaussian blur of testfile.m:57-58) plus Gaussian noise of sigma 1/255,
clipped to [0,1] -- the inverse-crime set-up of helpers/LoadImSequenceForward.m.
Everything is a pure function of (seed, frame index), so a rank can generate
its own frame block of a longer sequence.
"""
import numpy as np
import torch

GLOBAL_MOTION = (0.7, -0.4)      # per frame: (columns, rows)
DISC_MOTION = (-1.3, 0.9)


def gaussian_kernel(size=7, sigma=np.sqrt(0.6)):
    """fspecial('gaussian', size, sigma), testfile.m:58."""
    r = (size - 1) / 2.0
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    k = np.exp(-(x * x + y * y) / (2.0 * sigma * sigma))
    k[k < np.finfo(np.float64).eps * k.max()] = 0.0
    return k / k.sum()


def _scene(seed, Ny, Nx):
    rng = np.random.default_rng(seed)
    sc = {
        "patch_c": rng.random((24, 2)) * [Ny, Nx],
        "patch_sigma": rng.uniform(0.04, 0.12, 24) * min(Ny, Nx),
        "patch_theta": rng.uniform(0, np.pi, 24),
        "patch_freq": rng.uniform(0.02, 0.12, 24),
        "patch_amp": rng.uniform(0.05, 0.2, 24),
        "disc_c": rng.random((8, 2)) * [Ny, Nx],
        "disc_r": rng.uniform(0.03, 0.1, 8) * min(Ny, Nx),
        "disc_v": rng.uniform(-0.25, 0.25, 8),
        "mdisc_c": np.array([0.45 * Ny, 0.55 * Nx]),
        "mdisc_r": 0.08 * min(Ny, Nx),
        "mdisc_v": 0.3,
    }
    return sc


def render_frames(Ny, Nx, frames, seed=1234, device="cpu"):
    """HR ground truth g (Ny, Nx, len(frames)) float32 and the moving-disc masks."""
    sc = _scene(seed, Ny, Nx)
    dev = torch.device(device)
    r = torch.arange(Ny, dtype=torch.float32, device=dev)[:, None]
    c = torch.arange(Nx, dtype=torch.float32, device=dev)[None, :]
    out = torch.empty((Ny, Nx, len(frames)), dtype=torch.float32, device=dev)
    masks = torch.empty((Ny, Nx, len(frames)), dtype=torch.bool, device=dev)
    for i, t in enumerate(frames):
        y = r - GLOBAL_MOTION[1] * t          # static-scene coordinates of this frame's pixels
        x = c - GLOBAL_MOTION[0] * t
        g = 0.5 + 0.2 * (y / Ny - 0.5) + 0.1 * (x / Nx - 0.5)
        for k in range(24):
            dy = y - float(sc["patch_c"][k, 0])
            dx = x - float(sc["patch_c"][k, 1])
            s = float(sc["patch_sigma"][k])
            th = float(sc["patch_theta"][k])
            env = torch.exp(-(dy * dy + dx * dx) / (2 * s * s))
            g = g + float(sc["patch_amp"][k]) * env * torch.sin(
                2 * np.pi * float(sc["patch_freq"][k]) * (dy * np.cos(th) + dx * np.sin(th)))
        for k in range(8):
            dy = y - float(sc["disc_c"][k, 0])
            dx = x - float(sc["disc_c"][k, 1])
            d = torch.sqrt(dy * dy + dx * dx)
            g = g + float(sc["disc_v"][k]) * torch.sigmoid((float(sc["disc_r"][k]) - d) / 1.5)
        my = r - (float(sc["mdisc_c"][0]) + DISC_MOTION[1] * t)
        mx = c - (float(sc["mdisc_c"][1]) + DISC_MOTION[0] * t)
        md = torch.sqrt(my * my + mx * mx)
        g = g + sc["mdisc_v"] * torch.sigmoid((sc["mdisc_r"] - md) / 1.5)
        out[:, :, i] = torch.clamp(g, 0.0, 1.0)
        masks[:, :, i] = md < sc["mdisc_r"]
    return out, masks


def forward_model(g, factor=4, kernel=None, noise_seed=4321, frames=None, noise_sigma=1.0 / 255.0):
    """f = clip(D B g + noise): true convolution with replicate border, then block mean."""
    if kernel is None:
        kernel = gaussian_kernel()
    Ny, Nx, nf = g.shape
    k = torch.as_tensor(np.ascontiguousarray(kernel[::-1, ::-1]), dtype=torch.float32, device=g.device)
    kh, kw = k.shape
    x = g.permute(2, 0, 1)[:, None]                                     # (nf,1,Ny,Nx)
    x = torch.nn.functional.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2), mode="replicate")
    x = torch.nn.functional.conv2d(x, k[None, None])
    x = torch.nn.functional.avg_pool2d(x, factor)
    f = x[:, 0].permute(1, 2, 0)
    if noise_sigma > 0:
        frames = list(range(nf)) if frames is None else frames
        noise = np.stack([np.random.default_rng([noise_seed, int(t)]).normal(0.0, noise_sigma, f.shape[:2])
                          for t in frames], axis=2)
        f = f + torch.as_tensor(noise, dtype=torch.float32, device=g.device)
    return torch.clamp(f, 0.0, 1.0)


def flows_backward(masks):
    """v (Ny, Nx, npairs, 2) for pairs (t, t+1) of the frames `masks` was rendered for."""
    Ny, Nx, nf = masks.shape
    v = torch.empty((Ny, Nx, nf, 2), dtype=torch.float32, device=masks.device)
    v[..., 0] = torch.where(masks, DISC_MOTION[0], GLOBAL_MOTION[0])
    v[..., 1] = torch.where(masks, DISC_MOTION[1], GLOBAL_MOTION[1])
    return v


FLOW_KINDS = ("rigid", "smooth", "outliers")


def perturb_flows(v, pair_ids, kind, seed=777):
    """Irregular variants of the analytic flow, for throughput studies (the data term still comes
    from the rigid scene, so these flows do not describe it -- the sweeps do the same work).
      'smooth'   adds a low-frequency random field of about 3 px amplitude (bicubic blow-up of a
                 coarse 9x16 grid of N(0, 1.5) values), different for every pair;
      'outliers' additionally replaces 0.1 % of the vectors by uniform ones in [-200, 200]^2,
                 as an estimator does at occlusions.
    A pure function of (seed, global pair index): blocks of one sequence agree on shared pairs."""
    if kind == "rigid":
        return v
    if kind not in FLOW_KINDS:
        raise ValueError("flow kind must be one of %s" % (FLOW_KINDS,))
    Ny, Nx = v.shape[:2]
    out = v.clone()
    for i, q in enumerate(pair_ids):
        rng = np.random.default_rng([seed, int(q)])
        coarse = torch.as_tensor(rng.normal(0.0, 1.5, (1, 2, 9, 16)), dtype=torch.float32, device=v.device)
        field = torch.nn.functional.interpolate(coarse, size=(Ny, Nx), mode="bicubic", align_corners=True)[0]
        out[:, :, i, 0] += field[0]
        out[:, :, i, 1] += field[1]
        if kind == "outliers":
            cnt = max(int(round(1e-3 * Ny * Nx)), 1)
            idx = torch.as_tensor(rng.choice(Ny * Nx, size=cnt, replace=False), device=v.device)
            vec = torch.as_tensor(rng.uniform(-200.0, 200.0, (cnt, 2)), dtype=torch.float32, device=v.device)
            for comp in range(2):
                plane = out[:, :, i, comp].reshape(-1)          # a copy (the slice is not contiguous)
                plane[idx] = vec[:, comp]
                out[:, :, i, comp] = plane.reshape(Ny, Nx)
    return out


def make_block(ny, nx, nc_global, begin, end, factor=4, seed=1234, noise_seed=4321, device="cpu",
               with_truth=False, flow_kind="rigid"):
    """Inputs of the frame block [begin, end) of an nc_global-frame sequence.

    Returns numpy arrays: f (ny,nx,L), v for the pairs the block touches,
    the low-res frames needed for its start point (own + next block's first),
    and optionally the HR ground truth of the block.
    """
    Ny, Nx = ny * factor, nx * factor
    s_lo, s_hi = begin, min(end + 1, nc_global)
    p_lo, p_hi = (begin - 1 if begin > 0 else 0), min(end, nc_global - 1)
    frames = list(range(min(s_lo, p_lo), max(s_hi, p_hi + 1)))
    g, masks = render_frames(Ny, Nx, frames, seed, device)
    f_all = forward_model(g, factor, None, noise_seed, frames)
    off = frames[0]
    f_state = f_all[:, :, s_lo - off:s_hi - off]
    f_block = f_all[:, :, begin - off:end - off]
    v = flows_backward(masks[:, :, p_lo - off:max(p_hi, p_lo) - off])
    v = perturb_flows(v, range(p_lo, max(p_hi, p_lo)), flow_kind)
    out = {"f": f_block.cpu().numpy(), "f_state": f_state.cpu().numpy(), "v": v.cpu().numpy()}
    if with_truth:
        out["truth"] = g[:, :, begin - off:end - off].cpu().numpy()
    return out
