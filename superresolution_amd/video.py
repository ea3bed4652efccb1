"""Long videos as chained batches: the batching of
variants/VideoProcessor_supplementary_MMC.m:100-176 on top of the HIP solve.

Consecutive batches share one frame (`overlap = 1`, :24-32); the shared frame
of batch k+1 is pinned to the u and w that batch k produced for it
('u0_frame' / 'w0_frame', :144, :174-175), which the solver enforces with the
two identity duals of MultiframeMotionCoupling.m:535-572.  Flows are
precomputed for the whole video and sliced per batch.
"""
import numpy as np

from .mmc import MultiframeMotionCoupling


def batch_ranges(num_frames, batch_size, overlap=1):
    """0-based [begin, end) of every batch, as VideoProcessor computes them (:102-107, :125-143)."""
    if batch_size <= overlap:
        raise ValueError("batch size must exceed the overlap")
    step = batch_size - overlap
    count = int(np.ceil(num_frames / float(step)))
    batch_end = count * step - num_frames                        # :104
    if batch_end == 0:
        batch_end = batch_size                                   # :105-107
    out = []
    for kk in range(1, count + 1):
        if kk != count:
            b = (kk - 1) * step
            out.append((b, b + batch_size))                      # :127
        else:
            out.append((num_frames - batch_end, num_frames))     # :136
    return [(max(b, 0), min(e, num_frames)) for b, e in out if e > max(b, 0)]


def process_video(frames_lr, flows, batch_size, zoom_factor=4, k=None, configure=None, verbose=0, device=0):
    """Super-resolve a grey-level video ny x nx x T batch by batch.

    flows: Ny x Nx x (T-1) x 2.  `configure(mmc)` may set further properties
    (alpha, kappa, comp_mode, ...) on each batch object before init.
    Returns u, w (Ny x Nx x T each); frames covered by two batches keep the
    later batch's values, as the reference overwrites its output files.
    """
    frames_lr = np.asarray(frames_lr, dtype=np.float64)
    flows = np.asarray(flows, dtype=np.float64)
    ny, nx, T = frames_lr.shape
    Ny, Nx = ny * zoom_factor, nx * zoom_factor
    u_out = np.zeros((Ny, Nx, T))
    w_out = np.zeros((Ny, Nx, T))
    u0 = w0 = None                                               # :119-120
    done = 0
    for b, e in batch_ranges(T, batch_size):
        if done > b:
            # The pin is the result already computed for THIS batch's first frame.  For regular batches
            # that is the previous batch's last frame, as in the reference (:174-175); the reference's
            # shorter final batch (:136) starts earlier than that, where its rule would pin the wrong frame.
            u0, w0 = u_out[:, :, b].copy(), w_out[:, :, b].copy()
        m = MultiframeMotionCoupling(frames_lr[:, :, b:e], zoom_factor=zoom_factor,
                                     flowField=flows[:, :, b:e - 1, :] if e - b > 1 else np.zeros((Ny, Nx, 0, 2)),
                                     verbose=verbose, u0_frame=u0, w0_frame=w0, device=device)
        m.k = k
        if configure is not None:
            configure(m)
        m.init()
        m.run()
        u_out[:, :, b:e] = m._u
        w_out[:, :, b:e] = m._w
        m.close()
        done = e
    return u_out, w_out
