import sys, os, json
ROOT='/root/repo'
sys.path[:0]=[os.environ.get("GRAFT_REPO_ROOT", ROOT), os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
e = rtbhip.models.Panda().ets()
N=1000000
q = torch.from_numpy(np.random.default_rng(0).uniform(-3,3,(N,7))).cuda()
for mode in (0,1,0,1):
    rtbhip.tune("hess_mode", mode)
    f=lambda: e.hessian0(q)
    f(); ms,_,_=sustained_ms(f)
    print(json.dumps({"hess_mode":mode,"ms":round(ms,4),"frac":2408*N/(ms*1e-3)/8e12}))
rtbhip.tune("hess_mode",0)
