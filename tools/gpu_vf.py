"""View-factor workload on the GPU: the reference's cylinder examples, with timing.  Prints as it goes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as N
from tracer_amd.emissive_losses.view_factors_3D import Two_N_parameters_cavity_RTVF
N.set_printoptions(precision=4, suppress=True, linewidth=150)
t = time.time()
cyl = Two_N_parameters_cavity_RTVF(apertureRadius=1., frustaRadii=[1.], frustaDepths=[2.], coneDepth=0., el_FRUs=N.array([2]), el_CON=1,
                                   num_rays=int(float(sys.argv[1])) if len(sys.argv) > 1 else 400000, precision=0.002, seed=3, max_passes=20, verbose=True)
print('passes', cyl.passes, 'wall %.2f s' % (time.time() - t), flush=True)
print(cyl.VF_esperance, cyl.VF_esperance.sum(axis=1), flush=True)
t = time.time()
cav = Two_N_parameters_cavity_RTVF(1., [1.5, 1.5, 0.8], [0.5, 1.0, 0.4], 0.3, N.array([2, 3, 2]), 2, num_rays=200000, precision=0.003, seed=11, max_passes=20, verbose=True)
print('passes', cav.passes, 'wall %.2f s' % (time.time() - t), flush=True)
print(cav.VF_esperance, cav.VF_esperance.sum(axis=1), flush=True)
AF = cav.VF_esperance * N.vstack(cav.areas)
print('reciprocity max', N.abs(AF - AF.T).max(), flush=True)
