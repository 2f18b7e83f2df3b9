"""odin/rt_mi355x.odin against include/rt_mi355x.h, as text (no Odin compiler exists where this suite runs):
every procedure of the `foreign rt { ... }` block must be declared by the header with the same parameter names, in the
same order, with types of the same kind and width; rt_options must have the header's fields; every constant the binding
declares must carry the header's value, and the constants a host needs must all be there."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


# kind and width of a parameter, from either language
C_KINDS = [
    (r"^const char \*$", "cstring"), (r"^(const )?rt_options \*$", "ptr:rt_options"), (r"^(const )?rt_camera_render_data \*$", "ptr:camera_render_data"),
    (r"^(const )?rt_camera_settings \*$", "ptr:camera_settings"), (r"^(const )?rt_sphere \*$", "many:sphere"), (r"^(const )?rt_material \*$", "many:material"),
    (r"^(const )?rt_v3 \*$", "many:v3"), (r"^(const )?rt_camera_settings \*$", "ptr:camera_settings"), (r"^rt_world \*\*$", "ptr:rawptr"), (r"^rt_world \*$", "rawptr"), (r"^(const )?void \*$", "rawptr"), (r"^uint64_t \*$", "many:u64"),
    (r"^(const )?char \*$", "many:u8"), (r"^uint8_t \*$", "many:u8"), (r"^int64_t \*$", "ptr:i64"), (r"^rt_rect$", "rect"), (r"^rt_v3$", "v3"),
    (r"^int32_t$", "i32"), (r"^int64_t$", "i64"), (r"^uint32_t$", "u32"), (r"^uint64_t$", "u64"), (r"^double$", "f64"), (r"^void$", "void"),
]
ODIN_KINDS = {"cstring": "cstring", "^rt_options": "ptr:rt_options", "^camera_render_data": "ptr:camera_render_data", "^camera_settings": "ptr:camera_settings",
              "[^]sphere": "many:sphere", "[^]material": "many:material", "[^]v3": "many:v3", "^rawptr": "ptr:rawptr", "rawptr": "rawptr", "[^]u64": "many:u64",
              "[^]u8": "many:u8", "^i64": "ptr:i64", "rect": "rect", "v3": "v3", "i32": "i32", "i64": "i64", "u32": "u32", "u64": "u64", "f64": "f64"}


def c_kind(t):
    t = re.sub(r"\s+", " ", t.strip()).replace(" *", " *").replace("* *", "**")
    for pattern, kind in C_KINDS:
        if re.match(pattern, t):
            return kind
    raise AssertionError(f"unmapped C type {t!r}")


def header_api():
    text = strip_comments(open(os.path.join(ROOT, "include", "rt_mi355x.h")).read())
    functions = {}
    for ret, name, params in re.findall(r"\n\s*((?:const\s+)?\w+\s*\**)\s*\b(rt_\w+)\s*\(([^;{]*?)\)\s*;", text):
        plist = []
        for p in [q.strip() for q in params.replace("\n", " ").split(",")]:
            if p in ("void", ""):
                continue
            m = re.match(r"^(.*?)(\w+)$", p)
            ctype = re.sub(r"\s+", " ", m.group(1).strip())
            ctype = re.sub(r"\s*\*", " *", ctype).replace("* *", "**")
            plist.append((m.group(2), c_kind(ctype)))
        rtype = re.sub(r"\s*\*", " *", re.sub(r"\s+", " ", ret.strip()))
        functions[name] = (plist, c_kind(rtype))
    constants = {n: int(v.strip("()")) for n, v in re.findall(r"#define\s+(RT_\w+)\s+(\(?-?\d+\)?)", text)}
    struct = re.search(r"typedef struct \{([^}]*)\}\s*rt_options;", text).group(1)
    fields = []
    for line in struct.split(";"):
        line = line.strip()
        if line:
            m = re.match(r"^(.*?)(\w+)$", line)
            fields.append((m.group(2), c_kind(re.sub(r"\s*\*", " *", re.sub(r"\s+", " ", m.group(1).strip())))))
    return functions, constants, fields


def odin_params(text):
    out, pending = [], []
    for item in [q.strip() for q in text.replace("\n", " ").split(",") if q.strip()]:
        if ":" in item:
            name, typ = [x.strip() for x in item.split(":", 1)]
            for n in pending + [name]:
                out.append((n, ODIN_KINDS[typ]))
            pending = []
        else:
            pending.append(item)
    assert not pending, pending
    return out


def odin_api():
    text = strip_comments(open(os.path.join(ROOT, "odin", "rt_mi355x.odin")).read())
    block = re.search(r"foreign rt \{(.*?)\n\}", text, flags=re.S).group(1)
    functions = {}
    for name, params, ret in re.findall(r"(rt_\w+)\s*::\s*proc\((.*?)\)\s*(?:->\s*(\S+))?\s*---", block, flags=re.S):
        functions[name] = (odin_params(params), ODIN_KINDS[ret] if ret else "void")
    constants = {n: int(v) for n, v in re.findall(r"\n(RT_\w+)\s*::\s*(-?\d+)\b", text)}
    struct = re.search(r"rt_options :: struct \{(.*?)\n\}", text, flags=re.S).group(1)
    fields = odin_params(struct)
    return functions, constants, fields, text


def test_foreign_block_matches_the_header():
    c_fn, _, _ = header_api()
    o_fn, _, _, _ = odin_api()
    assert len(o_fn) >= 15
    for name, (params, ret) in o_fn.items():
        assert name in c_fn, f"{name} is bound in Odin but not declared in include/rt_mi355x.h"
        c_params, c_ret = c_fn[name]
        assert ret == c_ret, (name, ret, c_ret)
        assert [p[0] for p in params] == [p[0] for p in c_params], (name, "parameter names / order")
        assert [p[1] for p in params] == [p[1] for p in c_params], (name, "parameter types", params, c_params)
    # the entry points an Odin host needs
    for needed in ("rt_render_region", "rt_render_tiled", "rt_render_ppm", "rt_options_default", "rt_last_error", "rt_world_create", "rt_world_destroy",
                   "rt_render_accumulate", "rt_render_accumulate_static", "rt_finalize", "rt_finalize_ppm", "rt_release_cache", "rt_rng_u32"):
        assert needed in o_fn, needed


def test_options_struct_and_constants_match_the_header():
    _, c_const, c_fields = header_api()
    _, o_const, o_fields, text = odin_api()
    assert o_fields == c_fields
    for name, value in o_const.items():
        if name in ("RT_REJECTION_CAP",):
            continue                                  # the binding's own (the library's loop bound, restated)
        assert name in c_const, f"{name} is not a constant of the header"
        assert value == c_const[name], (name, value, c_const[name])
    for prefix in ("RT_PRECISION_", "RT_FETCH_", "RT_ERR_", "RT_STAT_", "RT_RNG_", "RT_FLAG_", "RT_MATERIAL_METALLIC_"):
        for name in c_const:
            if name.startswith(prefix):
                assert name in o_const, f"{name} is missing from odin/rt_mi355x.odin"
    assert o_const["RT_STATS_COUNT"] == c_const["RT_STATS_COUNT"] and o_const["RT_OK"] == 0
    # render_gpu takes what render_region takes (main.odin:510) plus the options a host may want to set
    sig = re.search(r"render_gpu :: proc\((.*?)\) -> bool", text, flags=re.S).group(1)
    for piece in ("image: image", "region: rect", "camera_render_data: camera_render_data", "world: world", "$print_progress: bool", "sphere_fetch: i32", "precision: i32"):
        assert piece in sig, piece


def test_rng_mirror_constants_are_those_of_the_library(rt):
    """the Odin mirror of the counter RNG uses the multipliers and shifts of the library's generator (rt_rng_u32)"""
    text = open(os.path.join(ROOT, "odin", "rt_mi355x.odin")).read()
    for literal in ("0x21f0aaad", "0x735a2d97", "0x9e3779b9", "0x85ebca6b", "0xc2b2ae35", "16777216.0"):
        assert literal in text, literal

    def mix32(x):
        x ^= x >> 16; x = (x * 0x21f0aaad) & 0xffffffff
        x ^= x >> 15; x = (x * 0x735a2d97) & 0xffffffff
        x ^= x >> 15
        return x

    def odin_draw(seed, pixel, sample, bounce, draw):   # rt_draws_open + (draw + 1) calls of rt_u32, as the Odin text reads
        k0 = mix32((seed & 0xffffffff) ^ 0x9e3779b9)
        k1 = mix32((seed >> 32) ^ k0 ^ 0x85ebca6b)
        kp, wp = mix32(k0 ^ pixel), mix32((k1 + pixel) & 0xffffffff)
        ks = mix32((kp + 0x9e3779b9 * (sample + 1)) & 0xffffffff)
        kb = mix32(ks ^ ((0x85ebca6b * (bounce + 1)) & 0xffffffff))
        c = (0xc2b2ae35 * (draw + 1)) & 0xffffffff
        return mix32((kb + c) & 0xffffffff) ^ wp

    for args in ((1, 0, 0, 0, 0), (1, 12345, 499, 3, 7), (0xdeadbeefcafef00d, 809999, 16383, 51, 63)):
        assert odin_draw(*args) == rt.rng_u32(*args)
