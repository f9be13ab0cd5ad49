import sys, numpy as np
sys.path.insert(0, '.')
import denseflow_amd as D
from denseflow_amd.synth import SynthClip
for (w,h) in [(1920,1080),(640,360),(1000,600)]:
    c = SynthClip(w,h,5); f0,f1 = c.frame(0), c.frame(2)
    with D.FlowEngine(w,h,"brox",impl=1,max_batch=2) as e: a = e.calc(f0,f1)
    with D.FlowEngine(w,h,"brox",max_batch=2) as e: b = e.calc(f0,f1); st=e.stats()
    d = np.abs(a-b).max(axis=2)
    ys,xs = np.nonzero(d>0)
    print(w,h,"levels",st.levels,"max diff",d.max(),"count",len(ys), "first", list(zip(xs[:5],ys[:5])), "x range",(xs.min(),xs.max()) if len(xs) else None, "y range",(ys.min(),ys.max()) if len(ys) else None, "finite", np.isfinite(b).all())
