"""rustracer_amd — MI355X (gfx950) render path for rustracer's raytracer_core, behind a C ABI.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of include/rthip.h),
`abi` (ctypes mirror of the header), `scene`/`raytracer` (host-side mirror of the reference's
scene/render API), `presets` (the benchmark scenes), `bands` (multi-GPU row-band scheduler).
"""
from . import abi  # noqa: F401
from .raytracer import (GenerationStatus, GroupRaytracer, PixelColor, PixelPosition, PixelRenderer, Raytracer, RenderResult,  # noqa: F401
                        host_array, make_params, owned_rows)
from .scene import (Camera, Dielectric, FlatScene, Lambertian, Mesh, Metal, Plane, PointLight, Sphere, SphereCloud,  # noqa: F401
                    Texture, Triangle, Vector3, flatten)
