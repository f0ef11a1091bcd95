import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, orc
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding
G='/root/repo/tests/golden'
for cfg,fi,fo in [((48000, 4800, 5000, 2, 2000, False),'lucky7.expected.cf32','lucky7.expected.nodc.s8'),((192000, 40000, 5000, 1, 2000, True),'nusat.cf32','processed.s8')]:
  iq=np.fromfile(os.path.join(G,fi),dtype=np.complex64); want=np.fromfile(os.path.join(G,fo),dtype=np.int8)
  for fast in (False, True):
      g=binding.Batch([cfg+(4096,)],keep_soft=True,fast_fma=fast)
      o8=[];of=[]
      for off in range(0,len(iq),4096):
          o8.append(g.process([iq[off:off+4096]])[0]); of.append(g.last_soft(0))
      o8=np.concatenate(o8); of=np.concatenate(of)
      d=np.abs(o8.astype(int)-want.astype(int))
      print("fast",fast,"len",len(o8),"nonzero",(d!=0).sum(),"max",d.max(),"where >2:",np.nonzero(d>2)[0][:20], "count>2",(d>2).sum())
      _,ex=orc.demod_stream(cfg,iq,4096)
      df=np.abs(of-ex)
      print("  float diff: max %.3g rms %.3g median %.3g; idx of 10 largest %s"%(df.max(), np.sqrt((df**2).mean()), np.median(df), np.argsort(df)[-10:]))
