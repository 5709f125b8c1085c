import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import centerface_amd as cfa
from centerface_amd import ops
from oracle import bf16_emulation as E
SD = cfa.weights.synthetic_state_dict(0)
prefix='layer1.1'
we, wd, wp = (SD["%s.conv.%s.weight" % (prefix, j)] for j in ("0.1", "1.1", "2"))
we=we.reshape(we.shape[0],-1); wp=wp.reshape(wp.shape[0],-1)
rng=np.random.default_rng(1)
x=E.q_bf16(torch.from_numpy((1.2*rng.standard_normal((1,24,160,160))).astype(np.float32))).numpy()
y=ops.mbconv(x,we,wd,wp,3,1,dtype="bf16")
ref=E.mbconv_fused(torch.from_numpy(x),we,wd,wp,3,1,True).numpy()
bad=~np.isfinite(y) | (np.abs(y-ref) > 0.05*np.abs(ref)+0.05)
print("bad frac", bad.mean(), "nan frac", (~np.isfinite(y)).mean())
b=bad[0]
print("by channel", b.mean(axis=(1,2)).round(3))
print("by y%16", np.array([b[:,i::16,:].mean() for i in range(16)]).round(3))
print("by x%16", np.array([b[:,:,i::16].mean() for i in range(16)]).round(3))
print("by tile x", np.array([b[:,:,i*16:(i+1)*16].mean() for i in range(10)]).round(3))
print("by tile y", np.array([b[:,i*16:(i+1)*16,:].mean() for i in range(10)]).round(3))
