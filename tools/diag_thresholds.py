import numpy as np, sys
sys.path.insert(0,'.')
from aind_smartspim_destripe_amd import engine as E, synth
g=np.load('tests/golden/large_stats.npz')
e=E.DestripeEngine(0)
for name,(h,w) in (('s1600x2000',(1600,2000)),('s2048',(2048,2048)),('s1800',(1800,1800))):
  for k in (0,1):
    img=synth.synthetic_plane(k,h,w)
    e.plan(h,w,synth.CELLS_CONFIG,synth.NO_CELLS_CONFIG,2500,max_batch=1)
    out=e.run(img[None],out_dtype=np.float32)
    key='%s__k%d__u16'%(name,k)
    ro=g[key+'__otsu'][::-1]; rt=g[key+'__thr'][::-1]
    for lv in range(e.levels):
        o,t=e.thresholds(0,lv)
        flag='' if abs(o-ro[lv])<=1e-4*ro[lv] else '   <<<<<< MISMATCH'
        print(key,lv,'otsu gpu %.6g ref %.6g  thr gpu %.6g ref %.6g'%(o,ro[lv],t,rt[lv]),flag)
