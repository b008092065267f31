import sys, numpy as np, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from concurrent.futures import ProcessPoolExecutor
import colate_amd
from colate_amd import workloads
def work(a):
    import oracle_lib as ol
    g,s,n,e=a
    return ol.em_batch(g,s,n,e)
import oracle_lib as ol
grid=ol.age_grid(); ep,_=ol.epochs_from_bins('3,7,0.2')
B=int(sys.argv[1])
csh,cns=workloads.sparse_tables(grid,B)
r1,it1,ll1,fl1=colate_amd.em_batch(grid,csh,cns,ep)
with ProcessPoolExecutor(14) as ex:
    res=list(ex.map(work,[(grid,csh[i:i+4],cns[i:i+4],ep) for i in range(0,B,4)]))
it0=np.concatenate([x[1] for x in res]); ll0=np.concatenate([x[2] for x in res]); fl0=np.concatenate([x[3] for x in res])
ok=(fl0&3)==0
bad=np.nonzero((it0!=it1)&ok)[0]
print(colate_amd.LIB_PATH, 'B',B,'mismatches',len(bad),'of',ok.sum(), [(int(b),int(it0[b]),int(it1[b]), float(ll1[b]/ll0[b]-1)) for b in bad[:10]], 'max |ll rel|', float(np.abs(ll1[ok]/ll0[ok]-1).max()), flush=True)
