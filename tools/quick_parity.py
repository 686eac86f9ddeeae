import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from hope_amd.capi import HopeLib, HopeContext
g = np.load('tests/golden/tum_example_depth.npz')
orc = HopeContext(HopeLib('oracle/_build/libhope_oracle.so','hope_oracle_'))
gpu = HopeContext(HopeLib())
for c in (orc,gpu):
    c.set_pose(g['t'], g['q']); st=c.segment_depth(g['depth']); k=c.counts(0)
    print(st, k.n_raw,k.n_vox,k.n_fit,k.n_segments,k.n_clusters,k.n_planes,list(k.grid),list(k.min_b), c.timing())
print('vox equal', np.array_equal(orc.downsampled().view(np.uint32), gpu.downsampled().view(np.uint32)))
a,b=orc.normals(),gpu.normals()
print('normals equal', np.array_equal(a.view(np.uint32),b.view(np.uint32)), np.nanmax(np.abs(a-b)))
print('fit equal', np.array_equal(orc.fit_indices(), gpu.fit_indices()))
print('knn equal', np.array_equal(orc.knn(), gpu.knn()))
print('labels equal', np.array_equal(orc.labels(), gpu.labels()))
ca,cb=orc.clusters(),gpu.clusters()
print('clusters equal', ca.tobytes()==cb.tobytes())
print(cb)
