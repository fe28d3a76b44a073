import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import oracle as O
from sdft_amd.sdft import SDFT
from sdft_amd.signals import noise, sine_sweep
def rel(a,b): return float(np.abs(np.asarray(a,dtype=np.float64)-b).max())/float(np.abs(b).max())
for combo,m,win,n in (("f32f32",3000,"hamming",2000),("f32f32",1024,"hann",6000),("f32f32",4096,"blackman",9000),("f64f32",1000,"hann",5000),("f32f64",1024,"hann",6000),("f32f64",2048,"blackman",6000)):
    td,fd,fdx=O.combo_types(combo)
    x=(sine_sweep(n,dtype=td)+noise(n,seed=2,dtype=td)*td(0.2))
    gain=np.linspace(1.0,0.0,m).astype(fd)
    for op in ("identity","gain"):
        ref=O.best(m,win,1.0,combo); d=ref.sdft(x)
        if op=="gain": d=(d*gain[None,:]).astype(fdx)
        want=ref.isdft(d)
        # exact-math reference: same pipeline in f64 types
        r64=O.best(m,win,1.0,"f64f64"); d64=r64.sdft(x.astype(np.float64))
        if op=="gain": d64=d64*gain[None,:].astype(np.float64)
        w64=r64.isdft(d64)
        out={}
        for fe,fold in ((0,1),(0,0),(2,0)):
            with SDFT(m,win,1.0,combo) as p:
                p.set_option("fused_exact",fe); p.set_option("fold",fold)
                out[(fe,fold)]=p.process(torch.from_numpy(x).cuda(),op,gain=gain).cpu().numpy()
        print(combo,m,win,op,"folded vs ref %.2e | tree(old) vs ref %.2e | ordered vs ref %.2e || vs f64 pipeline: ref %.2e folded %.2e"%(rel(out[(0,1)],want),rel(out[(0,0)],want),rel(out[(2,0)],want),rel(want,w64),rel(out[(0,1)],w64)),flush=True)
