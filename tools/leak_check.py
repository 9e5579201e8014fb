import sys, time, numpy as np
sys.path.insert(0, '.')
from mcmcpp_amd import capi, workloads
import torch
free0 = torch.cuda.mem_get_info()[0]
P = workloads.ar1_precision(32,0.5).ravel()
pos = workloads.init_positions(4096, 32, 1)
for rnd in range(5):
    for k in range(100):
        s = capi.HipSampler(4096, 32, capi.CALC_DENSE_GAUSSIAN, P, seed=k)
        s.set_state(pos, s.calc_logp(pos))
        s.run(3, interval=7)
        s.close()
    print("after %d cycles: device memory delta %d KiB" % ((rnd+1)*100, (free0-torch.cuda.mem_get_info()[0])//1024), flush=True)
for rnd in range(3):
    for k in range(100):
        s = capi.HipSampler(4096, 32, capi.CALC_DENSE_GAUSSIAN, P, seed=k, graph_steps=-1)
        s.set_state(pos, s.calc_logp(pos))
        s.run(3, interval=7, save_chain=False, want_accepted=False)
        s.close()
    print("no graphs/no chain, after %d more cycles: delta %d KiB" % ((rnd+1)*100, (free0-torch.cuda.mem_get_info()[0])//1024), flush=True)
