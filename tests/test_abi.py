"""The C-ABI library and the host-side helpers, without a GPU: the library loads, exports every
symbol the header declares, validates its arguments before touching a device, and its camera /
scene / PPM helpers agree with the oracle's restatement of the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    L = rt.lib()
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/rt_mi355x.h but not exported"
    assert sorted(rt.EXPORTED_SYMBOLS) == declared
    assert b"gfx950" in L.rt_version()


def test_struct_layouts_match_the_odin_structs(rt):
    assert C.sizeof(rt.V3) == 24
    assert rt.SPHERE_DTYPE.itemsize == 40 and rt.SPHERE_DTYPE.fields["radius"][1] == 24 and rt.SPHERE_DTYPE.fields["material_index"][1] == 32
    assert rt.MATERIAL_DTYPE.itemsize == 40 and rt.MATERIAL_DTYPE.fields["albedo"][1] == 8 and rt.MATERIAL_DTYPE.fields["param1"][1] == 32
    assert C.sizeof(rt.CameraRenderData) == 168 and rt.CameraRenderData.depth_of_field_angle.offset == 144
    assert C.sizeof(rt.CameraSettings) == 160 and C.sizeof(rt.Rect) == 32


def test_argument_validation_needs_no_device(rt):
    cs, sph, mat = rt.build_dev_scene(32)
    cam = rt.calculate_camera_render_data(cs)
    img = np.zeros((18, 32, 3))

    def code(fn):
        with pytest.raises(rt.RtError) as e:
            fn()
        assert e.value.message
        return e.value.code

    bad = sph.copy(); bad["material_index"][2] = 5
    assert code(lambda: rt.render_region(img, (0, 0, 32, 18), cam, bad, mat)) == rt.RT_ERR_INVALID_SCENE
    bad = sph.copy(); bad["radius"][1] = 0.0
    assert code(lambda: rt.render_region(img, (0, 0, 32, 18), cam, bad, mat)) == rt.RT_ERR_INVALID_SCENE
    badm = mat.copy(); badm["param1"][4] = 1.5     # metallic fuzz > 1: make_metallic_data asserts (main.odin:280)
    assert code(lambda: rt.render_region(img, (0, 0, 32, 18), cam, sph, badm)) == rt.RT_ERR_INVALID_SCENE
    badm = mat.copy(); badm["type"][0] = 7
    assert code(lambda: rt.render_region(img, (0, 0, 32, 18), cam, sph, badm)) == rt.RT_ERR_INVALID_SCENE
    assert code(lambda: rt.render_region(img, (0, 0, 33, 18), cam, sph, mat)) == rt.RT_ERR_INVALID_ARGUMENT   # region outside the image
    assert code(lambda: rt.render_region(img, (5, 0, 4, 18), cam, sph, mat)) == rt.RT_ERR_INVALID_ARGUMENT    # min > max
    cam0 = rt.calculate_camera_render_data(cs); cam0.samples_per_pixel = 0
    assert code(lambda: rt.render_region(img, (0, 0, 32, 18), cam0, sph, mat)) == rt.RT_ERR_INVALID_ARGUMENT
    assert code(lambda: rt.render_region(img, (0, 0, 32, 18), cam, sph, mat, precision=9)) == rt.RT_ERR_INVALID_ARGUMENT
    # NULL pointers straight through the ABI
    L = rt.lib()
    assert L.rt_render_region(None, 32, 18, rt.Rect(0, 0, 32, 18), C.byref(cam), sph.ctypes.data, len(sph), mat.ctypes.data, len(mat), None) == rt.RT_ERR_INVALID_ARGUMENT
    assert L.rt_world_create(sph.ctypes.data, len(sph), mat.ctypes.data, len(mat), 0, None) == rt.RT_ERR_INVALID_ARGUMENT
    assert L.rt_world_destroy(None) == rt.RT_OK
    # sizes are validated before anything is computed from them: no overflowed byte count is ever returned as a size
    huge = 1 << 40
    assert L.rt_render_ppm(huge, huge, C.byref(cam), sph.ctypes.data, len(sph), mat.ctypes.data, len(mat), None, None, 0) == rt.RT_ERR_UNSUPPORTED
    assert L.rt_render_ppm(1 << 20, 1 << 20, C.byref(cam), sph.ctypes.data, len(sph), mat.ctypes.data, len(mat), None, None, 0) == rt.RT_ERR_UNSUPPORTED
    assert L.rt_serialize_ppm(None, huge, huge, None, 0) == rt.RT_ERR_UNSUPPORTED
    assert L.rt_serialize_ppm(None, 1 << 15, 1 << 15, None, 0) == len(b"P3\n32768 32768\n255\n") + 12 * (1 << 30)   # the largest image: a size query
    assert L.rt_release_cache() == rt.RT_OK   # nothing cached, no device needed
    # an empty region is a successful no-op that needs no device either
    assert L.rt_render_region(img.ctypes.data, 32, 18, rt.Rect(3, 3, 3, 9), C.byref(cam), sph.ctypes.data, len(sph), mat.ctypes.data, len(mat), None) == rt.RT_OK
    assert np.all(img == 0)


def test_lookat_and_camera_match_the_oracle(rt, oracle):
    for pos, tgt in [((-2, 2, 1), (0, 0, -1)), ((13, 2, 3), (0, 0, 0)), ((0, 0, 0), (0, 0, -1))]:
        assert rt.lookat(pos, tgt, (0, 1, 0)) == oracle.lookat(pos, tgt, (0, 1, 0))
    for build in (lambda: rt.build_dev_scene(200), lambda: rt.build_dev_scene(400), lambda: rt.build_final_scene(1, 1200), lambda: rt.build_two_sphere_scene(400)):
        cs, _, _ = build()
        assert bytes(rt.calculate_camera_render_data(cs)) == bytes(oracle.calculate_camera_render_data(cs))
    # the dev scene's native 200-pixel width: 112.5 rows as a float, 112 as the image height (main.odin:729, :834)
    cs, _, _ = rt.build_dev_scene(200)
    assert cs.image_size_y == 112.5 and cs.image_height == 112


def test_camera_geometry(rt):
    cs, _, _ = rt.build_two_sphere_scene(400)
    cam = rt.calculate_camera_render_data(cs)
    # vfov 90 degrees at focus 1: the view plane is 2 high, 2*16/9 wide, centred on -z
    du, dv, tl = np.array(cam.pixel_delta_u.tuple()), np.array(cam.pixel_delta_v.tuple()), np.array(cam.view_plane_top_left_pixel_center.tuple())
    assert np.allclose(du * 400, (2 * 16 / 9, 0, 0)) and np.allclose(dv * 225, (0, -2, 0))
    centre = tl + du * 199.5 + dv * 112.0
    assert np.allclose(centre, (0, 0, -1))
    assert cam.depth_of_field_angle == 0.0 and np.allclose(cam.dof_disk_u.tuple(), 0)


def test_scene_builders(rt):
    cs, sph, mat = rt.build_dev_scene(200)
    assert len(sph) == 5 and len(mat) == 5 and cs.samples_per_pixel == 100 and cs.max_ray_bounces == 50
    assert list(mat["type"]) == [0, 0, 2, 2, 1] and mat["param1"][3] == 1.0 / 1.5 and tuple(sph["center"][0]) == (0.0, -100.5, -1.0)
    cs, sph, mat = rt.build_final_scene(1, 1200)
    assert (cs.image_width, cs.image_height) == (1200, 675) and cs.samples_per_pixel == 500
    assert 470 <= len(sph) <= 488 and len(mat) == len(sph)       # 4 + up to 484, one material per sphere
    small = sph[4:]
    assert np.all(small["radius"] == 0.2) and np.all(small["center"][:, 1] == 0.2)
    assert np.all(np.linalg.norm(small["center"] - np.array((4, 0.2, 0)), axis=1) > 0.9)
    kinds = np.bincount(mat["type"][4:], minlength=3) / len(small)
    assert 0.7 < kinds[0] < 0.9 and 0.08 < kinds[1] < 0.22 and 0.01 < kinds[2] < 0.1
    metal = mat[4:][mat["type"][4:] == 1]
    assert np.all((metal["albedo"] >= 0.5) & (metal["albedo"] < 1)) and np.all((metal["param1"] >= 0) & (metal["param1"] < 0.5))
    a = rt.build_final_scene(1, 1200)[1]; b = rt.build_final_scene(2, 1200)[1]
    assert np.array_equal(a, sph) and not np.array_equal(a["center"][4:40], b["center"][4:40])
    cs, sph, mat = rt.build_stress_scene(5, 10000, 1920)
    assert len(sph) == 10000 and (cs.image_width, cs.image_height) == (1920, 1080) and cs.samples_per_pixel == 256
    assert np.all((sph["radius"][1:] >= 0.05) & (sph["radius"][1:] < 0.2)) and np.all(np.abs(sph["center"][1:, [0, 2]]) <= 30)
    # capacity handling through the raw ABI
    L = rt.lib()
    s = rt.CameraSettings(); ns, nm = C.c_int64(2), C.c_int64(2)
    two = np.zeros(2, dtype=rt.SPHERE_DTYPE); twom = np.zeros(2, dtype=rt.MATERIAL_DTYPE)
    assert L.rt_build_dev_scene(200, C.byref(s), two.ctypes.data, C.byref(ns), twom.ctypes.data, C.byref(nm)) == rt.RT_ERR_BUFFER_TOO_SMALL
    assert (ns.value, nm.value) == (5, 5)


def test_ppm_writer_matches_the_oracle(rt, oracle):
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 1.2, size=(7, 5, 3))
    img[0, 0] = (0.0, 0.999, 1.0); img[0, 1] = (0.0390625, 0.00390624, 0.5)   # 10/256 exactly, just under 1/256
    ppm = rt.serialize_ppm(img)
    assert ppm == oracle.serialize_ppm(img)
    assert ppm.startswith(b"P3\n5 7\n255\n") and len(ppm) == len(b"P3\n5 7\n255\n") + 12 * 35
    lines = ppm.split(b"\n")
    assert lines[3] == b"  0 255 255" and lines[4] == b" 10   0 128"
    assert np.array_equal(rt.quantize_rgb8(img), oracle.quantize_rgb8(img))
    assert rt.quantize_rgb8(img).max() == 255


def test_bvh_builder_invariants(rt):
    """the hierarchy of RT_FETCH_BVH, built and checked on the host for all eight traversal orders"""
    import numpy as np
    for build in (lambda: rt.build_two_sphere_scene(400), lambda: rt.build_dev_scene(400), lambda: rt.build_final_scene(1, 1200),
                  lambda: rt.build_final_scene(7, 1200), lambda: rt.build_stress_scene(5, 10000, 1920)):
        _, sph, _ = build()
        n_nodes, max_leaf = rt.bvh_validate(sph)
        assert 1 <= max_leaf <= 4 and len(sph) / 4 <= n_nodes <= 2 * len(sph)
    same = np.zeros(37, dtype=rt.SPHERE_DTYPE)
    same["radius"] = 0.5                                  # 37 coincident spheres: no split separates them
    n_nodes, max_leaf = rt.bvh_validate(same)
    assert max_leaf <= 4 and n_nodes >= 10
    assert rt.bvh_validate(same[:0]) == (0, 0)
    assert rt.bvh_validate(same[:1]) == (1, 1)
