import sys, time, numpy as np
sys.path.insert(0, '.')
from mcmcpp_amd import capi, workloads
import torch
t0=time.time()
free0 = torch.cuda.mem_get_info()[0]
for k in range(150):
    W, D = [(64,4),(4096,32),(1000,9),(16384,32)][k%4]
    calc = [capi.CALC_ISO_GAUSSIAN, capi.CALC_DENSE_GAUSSIAN, capi.CALC_ROSENBROCK, capi.CALC_DENSE_GAUSSIAN][k%4]
    prm = [None, workloads.ar1_precision(32,0.5).ravel(), [1.0,100.0,0.05], workloads.ar1_precision(32,0.5).ravel()][k%4]
    s = capi.HipSampler(W, D, calc, prm, seed=k)
    pos = workloads.init_positions(W, D, k)
    s.set_state(pos, s.calc_logp(pos))
    s.run(3, interval=7)
    s.close()
free1 = torch.cuda.mem_get_info()[0]
print("150 create/run/destroy cycles: %.1f s, device memory delta %d KiB" % (time.time()-t0, (free0-free1)//1024))
s = capi.HipSampler(16384, 32, capi.CALC_DENSE_GAUSSIAN, workloads.ar1_precision(32,0.5).ravel(), seed=1)
pos = workloads.init_positions(16384, 32); s.set_state(pos, s.calc_logp(pos))
t0=time.time(); chain, acc = s.run(20, interval=10000, want_accepted=True); dt=time.time()-t0
c = s.counters()
print("200000 steps in %.2f s (%.3e w-s/s), acceptance %.4f, near_ties %d, steps %d" % (dt, 16384*200000/dt, acc.mean()/16384, c['near_ties'], c['ensemble_steps']))
cov = np.cov(chain[5:].reshape(-1,32).T); sigma = 0.5**np.abs(np.subtract.outer(np.arange(32),np.arange(32)))
print("max |cov - Sigma| over last 15 stored steps:", np.abs(cov-sigma).max())
