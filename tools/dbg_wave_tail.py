import sys
sys.path.insert(0, "/root/repo")
from rustracer_amd import Raytracer, abi, flatten, presets
flat = flatten(presets.standalone_scene())
rt = Raytracer.new(3840, 2160, seed=1)
for spp in (64, 256):
    s = rt.render(flat, spp, max_depth=5, want_accum=False, flags=abi.F_COUNT_STATS).stats
    print("spp", spp, "rays", s["rays"], "wave-iterations*64", s["shadow_rays"], "lanes busy per iteration %.3f" % (s["rays"] / s["shadow_rays"]))
