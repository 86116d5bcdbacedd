"""Host-side cost of a procedural hall: generation and akr_scene_create without a context (flattening, records, light tables, BVH build).
usage: [AKR_TIMING=1] [AKR_HOST_THREADS=n] python tools/hall_compile_time.py <triangles>"""
import time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from akari_render_amd import procedural, capi
n=int(sys.argv[1])
t0=time.time(); sd=procedural.sponza_like(n, seed=1234, width=1920, height=1080); t1=time.time()
sc=capi.Scene(None, sd); t2=time.time()
info=sc.info()
print("tris",info.n_triangles,"nodes",info.n_bvh_nodes,"generate %.2f s compile %.2f s"%(t1-t0,t2-t1))
