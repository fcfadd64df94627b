import importlib, sys, collections
sys.path.insert(0,'/root/repo'); 
import os
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,ROOT)
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
d = za.silesia_mix(64<<20).tobytes()
out,_ = enc.deflate(d,10)
b = enc.last_blocks()
fm = collections.Counter(int(x) for x in b[:,2])
print('blocks',len(b),'fmt hist',dict(fm))
# recycle chain lengths
chains=collections.Counter(); run=0
for f in b[:,2]:
    if int(f)==4: run+=1
    else:
        if run: chains[run]+=1
        run=0
if run: chains[run]+=1
print('recycle chain lengths',sorted(chains.items())[:20], 'max', max(chains) if chains else 0)
import numpy as np
print('atoms/block mean', b[:,1].mean(), 'median', np.median(b[:,1]))
