import os, sys
sys.path.insert(0, "/root/repo")
from rustracer_amd import Raytracer, abi, flatten, presets
for name, scene, w, h, spp, depth in (("C4", presets.mesh_scene, 3840, 2160, 4, 50), ("C5", lambda: presets.big_scene(100, 7, "uv100k"), 7680, 4320, 8, 8)):
    flat = flatten(scene())
    rt = Raytracer.new(w, h, seed=1)
    s = rt.render(flat, spp, max_depth=depth, want_accum=False, flags=abi.F_COUNT_STATS).stats
    print(name, "inner: wave-steps*64", s["sky_evals"], "lane-steps", s["shadow_rays"], "util %.3f" % (s["shadow_rays"] / max(1, s["sky_evals"])),
          "| leaf: wave-iters*64", s["hits"][0], "lane-iters", s["hits"][1], "util %.3f" % (s["hits"][1] / max(1, s["hits"][0])),
          "| rays", s["rays"], "inner lane-steps/ray %.1f" % (s["shadow_rays"] / s["rays"]), "leaf lane-iters/ray %.1f" % (s["hits"][1] / s["rays"]),
          "inner wave-steps/ray(x64) %.1f" % (s["sky_evals"] / s["rays"]), "leaf %.1f" % (s["hits"][0] / s["rays"]))
    rt.close()
