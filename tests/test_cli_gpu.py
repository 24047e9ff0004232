"""The C++ front-end (rustracer_amd/csrc/rthip_cli, built on raytracer_hip.hpp) drives the same C ABI as the Python
layer: its PPM must be byte-identical to the Python render of the same scene/seed, and it exercises the
PixelRenderer::set_pixel hand-off and the PPM writer (row order of standalone/src/renderers/ppm.rs:28-35)."""
import json
import os
import subprocess

import numpy as np
import pytest

from rustracer_amd import Dielectric, Lambertian, Metal, Raytracer, Sphere, Triangle, presets

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "rustracer_amd", "csrc", "rthip_cli")


def test_cli_matches_python_layer(tmp_path):
    assert os.path.exists(CLI), "build first: __graft_entry__.build()"
    w, h, spp, depth, seed = 200, 112, 3, 50, 9
    out = tmp_path / "img.ppm"
    r = subprocess.run([CLI, "--width", str(w), "--height", str(h), "--spp", str(spp), "--depth", str(depth), "--seed", str(seed),
                        "--texture", presets.texture_path("bergsjostolen"), "--out", str(out)], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    assert info["paths"] == w * h * spp and info["set_pixel_calls"] == w * h and info["shape_at_centre"] == 3
    tok = out.read_text().split()
    assert tok[0] == "P3" and (int(tok[1]), int(tok[2]), int(tok[3])) == (w, h, 255)
    ppm = np.array(tok[4:], dtype=np.uint8).reshape(h, w, 3)
    rt = Raytracer.new(w, h, seed=seed)
    res = rt.render(presets.standalone_scene(), spp, max_depth=depth)
    assert info["rays"] == res.stats["rays"]
    assert np.array_equal(ppm, res.rgba8[::-1, :, :3])  # the PPM is written top row first
    rt.close()


def test_cli_progressive_adaptive_passes(tmp_path):
    """--passes: the standalone window's loop with the reference's convergence rule; set_pixel only for pixels that were not
    Final before a pass, each Final pixel reported as Final exactly once (lib.rs:277-279,309-314)"""
    w, h, passes, depth, seed = 160, 90, 10, 50, 4
    out = tmp_path / "img.ppm"
    r = subprocess.run([CLI, "--width", str(w), "--height", str(h), "--spp", "1", "--depth", str(depth), "--seed", str(seed), "--passes", str(passes),
                        "--texture", presets.texture_path("bergsjostolen"), "--out", str(out)], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    rt = Raytracer.new(w, h, seed=seed)
    paths = 0
    scene = presets.standalone_scene()   # one object: a new scene would be re-uploaded, which invalidates the running sum
    for k in range(passes):
        res = rt.render(scene, 1, max_depth=depth, accumulate=k > 0, adaptive=True)
        paths += res.stats["paths"]
    assert info["paths"] == paths == info["set_pixel_calls"]   # spp = 1: one set_pixel per traced pixel
    assert info["final_pixels"] == res.stats["pixels_final"] == info["final_set_pixel_calls"] > 0
    tok = out.read_text().split()
    ppm = np.array(tok[4:], dtype=np.uint8).reshape(h, w, 3)
    assert np.array_equal(ppm, res.rgba8[::-1, :, :3])
    rt.close()


def _grid_scene(n):
    """the scene of `rthip_cli --scene grid --grid n` (rthip_cli.cpp: grid_scene), through the Python constructors"""
    s = [Sphere.new((0.0, -100.5, -1.0), 100.0, Lambertian.new_from_hex(0x007070))]
    for i in range(n):
        for j in range(n):
            c = ((i - n * 0.5) * 0.6, -0.3 + 0.05 * ((i * 7 + j * 3) % 5), -1.0 - j * 0.6)
            radius = 0.1 + 0.02 * ((i + 2 * j) % 4)
            k = (i + j) % 3
            if k == 0:
                m = Lambertian.new((0.2 + 0.1 * (i % 5), 0.5, 0.3 + 0.1 * (j % 4)))
            elif k == 1:
                m = Metal.new((0.8, 0.8, 0.8), 0.1 * ((i + j) % 4))
            else:
                m = Dielectric.new((1.0, 1.0, 1.0), 1.5)
            s.append(Sphere.new(c, radius, m))
    wall = Lambertian.new((0.7, 0.6, 0.2))
    z = -1.5 - n * 0.6
    s.append(Triangle.new((-4.0, -0.5, z), (4.0, -0.5, z), (4.0, 2.0, z), wall))
    s.append(Triangle.new((-4.0, -0.5, z), (4.0, 2.0, z), (-4.0, 2.0, z), wall))
    return s


@pytest.mark.parametrize("name,grid", [("wasm", 0), ("grid", 4), ("grid", 14)])
def test_cli_scene_presets(tmp_path, name, grid):
    """the other scenes of the C++ front-end: wasm/src/lib.rs:90-105, and sphere grids with a triangle quad (grid 4: LDS scan,
    grid 14: 197 spheres + 2 triangles through the BVH) — byte-identical to the same scene built with the Python constructors"""
    w, h, spp, depth, seed = 160, 90, 2, 8, 3
    out = tmp_path / "img.ppm"
    r = subprocess.run([CLI, "--scene", name, "--grid", str(max(grid, 1)), "--width", str(w), "--height", str(h), "--spp", str(spp), "--depth", str(depth),
                        "--seed", str(seed), "--out", str(out)], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    scene = presets.wasm_scene() if name == "wasm" else _grid_scene(grid)
    rt = Raytracer.new(w, h, seed=seed)
    res = rt.render(scene, spp, max_depth=depth)
    assert info["rays"] == res.stats["rays"] and info["paths"] == w * h * spp
    tok = out.read_text().split()
    ppm = np.array(tok[4:], dtype=np.uint8).reshape(h, w, 3)
    assert np.array_equal(ppm, res.rgba8[::-1, :, :3])
    rt.close()


def test_cli_rgba_out_is_the_top_down_frame(tmp_path):
    """--rgba-out: the raw RGBA8 frame in SCREEN order, as standalone's pixels front-end draws it
    (standalone/src/renderers/pixels.rs:240-260: screen row 0 shows pos.y = height-1) == the PPM rows (written top-down,
    ppm.rs:28-35) + alpha 255 == the y-up buffer of the C ABI flipped"""
    w, h, spp, seed = 176, 98, 2, 5
    ppm_path, rgba_path = tmp_path / "img.ppm", tmp_path / "frame.rgba"
    r = subprocess.run([CLI, "--width", str(w), "--height", str(h), "--spp", str(spp), "--depth", "6", "--seed", str(seed),
                        "--texture", presets.texture_path("bergsjostolen"), "--out", str(ppm_path), "--rgba-out", str(rgba_path)],
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    frame = np.fromfile(rgba_path, dtype=np.uint8).reshape(h, w, 4)
    tok = ppm_path.read_text().split()
    ppm = np.array(tok[4:], dtype=np.uint8).reshape(h, w, 3)
    assert np.array_equal(frame[..., :3], ppm) and (frame[..., 3] == 255).all()
    rt = Raytracer.new(w, h, seed=seed)
    res = rt.render(presets.standalone_scene(), spp, max_depth=6)
    assert np.array_equal(frame, res.rgba8[::-1])
    rt.close()


def test_cli_devices_group_matches_single_device(tmp_path):
    """--devices: the C++ host layer through rthip_group_* (a 3-way band split rehearsed on one GPU with peer copies)"""
    w, h, spp, seed = 200, 117, 3, 8
    outs = []
    for extra in ([], ["--devices", "0"], ["--devices", "0,0,0", "--peer-copy", "1"]):
        out = tmp_path / f"img{len(outs)}.ppm"
        r = subprocess.run([CLI, "--width", str(w), "--height", str(h), "--spp", str(spp), "--depth", "6", "--seed", str(seed),
                            "--texture", presets.texture_path("bergsjostolen"), "--out", str(out)] + extra,
                           capture_output=True, text=True, timeout=120, cwd=ROOT)
        assert r.returncode == 0, r.stderr
        info = json.loads(r.stdout)
        assert info["paths"] == w * h * spp and info["set_pixel_calls"] == w * h
        outs.append((out.read_bytes(), info))
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert outs[0][1]["rays"] == outs[1][1]["rays"] == outs[2][1]["rays"] and outs[2][1]["devices"] == 3
