import json,sys,os
sys.path.insert(0,os.getcwd())
from hagrid_amd import api, scene
mem=api.MemManager(keep=True); n=1000000
tris=scene.make_soup(n); d=mem.upload(tris)
g=api.build_all(mem,d,n)
bc=mem.build_counts()
print(json.dumps({k:bc[k] for k in ("level_refs","level_cells","level_kept","build_cells","build_refs","build_entries","merge_cells","merge_refs","merged_cells","merged_refs","flatten_entries_in","flatten_entries_out","top_refs","top_cells")}))
