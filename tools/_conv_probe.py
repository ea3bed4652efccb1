import sys, os, numpy as np
root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,'oracle')); sys.path.insert(0, os.path.join(root,'tests','golden'))
import mmc_oracle as orc
import make_converged as mc
from superresolution_amd.solver import FrameBlockSolver
truth, lr, v = mc.tube3_center()
k = orc.fspecial_gaussian(7, np.sqrt(0.6))
for kappa in (0.25, float('nan')):
    ny,nx,nc = lr.shape
    with FrameBlockSolver(ny,nx,4,nc,kappa=kappa) as s:
        s.set_lowres(lr); s.set_blur(k); s.set_flow(v); s.set_state(); s.init()
        done=0
        for target in (800,2500,7500,20000,50000,100000,200000,400000):
            s.solve(target-done,0.0,0); done=target
            d=s.diagnostics(); u,w=s.get_state()
            print(kappa, target, 'E %.9f dual %.9f infeas %.3e psnr %.4f' % (d['primal_energy'], d['dual_energy'], d['dual_infeasibility'], orc.psnr_central(u.astype(np.float64), truth)), flush=True)
