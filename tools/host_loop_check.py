"""Host-side cost of the sharded iteration loop: sharding.solve_sharded on a single block (no
neighbours, so no transfers) against the library's own loop mmc_solve, same workload."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from superresolution_amd import sharding, synthetic
from superresolution_amd.imresize import upsample_sequence
from superresolution_amd.solver import FrameBlockSolver

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
L, ny, nx = int(sys.argv[1]) if len(sys.argv) > 1 else 30, 270, 480
blk = synthetic.make_block(ny, nx, L, 0, L, device="cuda:0")
u_up = upsample_sequence(blk["f_state"], 4).astype(np.float32)
s = FrameBlockSolver(ny, nx, 4, L, stream=torch.cuda.current_stream().cuda_stream)
s.set_lowres(blk["f"]); s.set_blur(synthetic.gaussian_kernel()); s.set_flow(blk["v"]); s.set_state(u_up, u_up); s.init()
ex = sharding.HaloExchanger(sharding.halo_tensors(s), 0, 1)
for name, fn in (("mmc_solve", lambda n: s.solve(n, 0.0, 0)),
                 ("solve_sharded", lambda n: sharding.solve_sharded(s, ex, n, 0.0, 0, "cuda:0"))):
    fn(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(200)
    torch.cuda.synchronize()
    print("%-14s %.4f ms per iteration" % (name, (time.perf_counter() - t0) / 200 * 1e3), flush=True)
# host time alone: enqueue without waiting for the GPU
t0 = time.perf_counter()
sharding.solve_sharded(s, ex, 200, 0.0, 0, "cuda:0")
print("solve_sharded includes its final sync; host enqueue rate is bounded by the above")
s.close()
dist.destroy_process_group()
