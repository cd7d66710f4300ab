import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model as om
P=dict(q_lo=.05,q_hi=.95,q_lo_weight=1,q_hi_weight=1,mse_weight=1)
print("cores", os.cpu_count())
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    st = om.det_state(1,1)
    x=torch.randn(4,1,320,320); y=torch.rand(4,1,320,320)
    om.train_steps(st,[(x,y)],P,1e-4)
    t0=time.perf_counter(); om.train_steps(st,[(x,y)],P,1e-4); dt=time.perf_counter()-t0
    print(f"threads {th}: {4/dt:.3f} img/s ({dt:.1f}s/step)", flush=True)
