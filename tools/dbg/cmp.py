import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
if a.shape != b.shape:
    h,w=b.shape[0],b.shape[1]; a=a.reshape(h,a.shape[0]//h,w,a.shape[1]//w,3).mean(axis=(1,3))
d=np.abs(a-b).max(axis=2); ys,xs=np.nonzero(d>2e-6)
print(sys.argv[1], sys.argv[2], a.shape, 'differing pixels', len(ys), 'max', d.max(), list(zip(xs[:12].tolist(),ys[:12].tolist())))
