"""DEV TOOL: what the context remembers about a ray buffer (hagrid_kat_order_state) over bursts of ten launches: the trials of traverse.hip at work.
usage: python tools/dev_order_state.py [scene] [WxH,...] [bursts]"""
import sys, json, numpy as np
sys.path.insert(0,'/root/repo')
from hagrid_amd import api, scene
mem=api.MemManager(keep=True)
SC=sys.argv[1] if len(sys.argv)>1 else 'clustered'
tris=scene.make_soup(1000000) if SC=='soup' else getattr(scene,'make_'+SC)(); d_tris=mem.upload(tris)
grid=api.build_all(mem,d_tris,tris.shape[0]); api.setup_traversal(grid)
for (w,h) in [tuple(int(v) for v in a.split('x')) for a in (sys.argv[2] if len(sys.argv)>2 else '1024x1024').split(',')]:
    rays=scene.make_rays_primary(grid.bbox_min,grid.bbox_max,w,h); n=rays.shape[0]
    d_rays=mem.upload(rays); d_hits=mem.alloc(16*n)
    for burst in range(int(sys.argv[3]) if len(sys.argv)>3 else 16):
        for _ in range(10): api.traverse_grid(grid,d_tris,d_rays,d_hits,n)
        mem.synchronize()
        if True:
            ms=api.profile(lambda: [api.traverse_grid(grid,d_tris,d_rays,d_hits,n) for _ in range(10)], mem)/10
            print(w,h,"launches",(burst+1)*10+ (burst//2+1)*10,"ms %.4f"%ms, json.dumps({k:v for k,v in mem.order_state(d_rays).items() if k in ("head_tiles","head_dropped","n_base","n_head","n_all","learned_all","head_suggested","share_choice","order_loses","ms_base","ms_head","ms_all","ms_share_best")}), flush=True)
    mem.free(d_rays); mem.free(d_hits)
