#!/usr/bin/env python3
"""scripts/gpu_fuzz_paths.py -- robot.ets(start=, end=) / robot.fkine(q, end=, start=) of random branched robots (numbered automatically and by hand) on the
device, handed the ROBOT's q: every path -- descending, and climbing through inverted links -- against the product of the links' own transforms.
Exit code 1 on a miss."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path[:0] = [os.path.join(ROOT, 'tests'), ROOT, os.path.join(ROOT, 'robotics-toolbox-python_amd')]
import numpy as np
import rtbhip
from rtbhip import ERobot
from oracle import chains
from test_erobot_rne import random_tree
from test_erobot_dynamics import renumbered_case
def link_T(l, q, jidx):
    T=np.eye(4)
    for it in l["ets"]:
        if isinstance(it, np.ndarray): T=T@it
        elif len(it)>1 and it[1] is not None: T=T@chains.elementary(it[0], it[1])
        else: T=T@chains.elementary(it[0], -q[jidx] if (len(it)>2 and it[2]) else q[jidx])
    return T
bad=0; ran=0
if True:
    for seed in range(40):
        rng=np.random.default_rng(900+seed)
        if seed%2==0:
            prod, orc = random_tree(rng, n_links=int(rng.integers(3,12))); rob=ERobot(prod)
        else:
            try: rob, orc, rng = renumbered_case(900+seed, 4+seed%7)
            except Exception: continue
        n=rob.n
        if n==0: continue
        byname={l["name"]:l for l in orc}; jix={l.name:l.jindex for l in rob.links}
        def world(name, qrow):
            if name is None: return np.eye(4)
            l=byname[name]; return world(l["parent"],qrow) @ link_T(l,qrow,jix[name])
        def anc(name):
            out=[]
            while name is not None: out.append(name); name=byname[name]["parent"]
            return out
        q=rng.uniform(-2,2,(3,n))
        names=[l.name for l in rob.links]
        for _ in range(12):
            a,b=rng.choice(names),rng.choice(names)
            try:
                e=rob.ets(start=a,end=b)
            except ValueError as ex:
                continue
            if a in anc(b): want=[np.linalg.inv(world(byname[a]["parent"],r))@world(b,r) for r in q]
            else: want=[np.linalg.inv(world(a,r))@world(b,r) for r in q]
            try:
                got=np.asarray(e.eval(q)); d=np.abs(got-np.array(want)).max()
                J=e.jacob0(q); H=e.hessian0(q)
                got2=np.asarray(rob.fkine(q,end=b,start=a,include_base=False)) if hasattr(rob,'fkine') else got
                d2=np.abs(np.asarray([getattr(g,'A',g) for g in got2]).reshape(-1,4,4)-np.array(want)).max()
            except Exception as ex:
                print("FAILED",seed,a,b,repr(ex)[:160]); bad+=1; continue
            ran+=1
            if not (d<1e-10 and d2<1e-10): bad+=1; print("MISMATCH",seed,a,b,d,d2)
print(json.dumps({"paths": ran, "misses": bad}))
sys.exit(1 if bad else 0)
