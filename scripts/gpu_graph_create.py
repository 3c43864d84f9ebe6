import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import DeviceGraph, synth
t=time.time(); ia, ja = synth.random_graph_csr(1000000, 4500000); print("numpy CSR build %.2f s" % (time.time()-t))
for k in range(2):
    t=time.time(); g = DeviceGraph(ia, ja); print("graph_create (with edge index) %.2f s" % (time.time()-t)); g.close()
t=time.time(); g = DeviceGraph(ia, ja, n_edge_cols=0); print("graph_create (no edge ids) %.2f s" % (time.time()-t))
