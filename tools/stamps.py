#!/usr/bin/env python3
"""Per-phase cycle shares of the megakernel from a -DPT_STAMPS diagnostic build (PT_LIB_PATH)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cudapathtracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = sys.argv[2] if len(sys.argv) > 2 else "cornell"
tmp = tempfile.mkdtemp()
s = getattr(scenes, wl)(tmp, width=1920, height=1080, spp=spp, max_depth=8)
hs = api.HostScene(s["config"]); sc = api.Scene(hs)
tiles = torch.zeros(api.n_tiles(1920, 1080), 64, 4, device="cuda")
sc.reset_counters()
sc.render_tiles_device(hs.camera(), 1920, 1080, spp, 8, tiles.data_ptr(), count_work=True)
torch.cuda.synchronize()
st = sc.debug_stamps()
life, nstart, end = st.pop("wave_lifetimes"), st.pop("not_earliest_start"), st.pop("latest_end")
span = end - ((~nstart) & (2**64 - 1))
if span > 0:
    slots = 256 * 16
    print("wave lifetimes %.3f s-slots, span %.3f ms, mean resident waves %.0f of %d (%.1f %%)" %
          (life / 1e8, span / 1e5, life / span, slots, 100.0 * life / span / slots))
tot = sum(st.values()) or 1
print("kernel ms", sc.last_kernel_ms())
for k, v in st.items(): print("%-11s %6.2f %%" % (k, 100.0 * v / tot))
print(sc.counters())
