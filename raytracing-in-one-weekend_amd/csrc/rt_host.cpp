// Host-side procedures on either side of the render call: camera set-up, the scene builders
// and the PPM writer.  They stay on the CPU in the reference (main.odin:65-71, :450-485,
// :687-789, :791-824) and stay on the CPU here; the C ABI exposes them so that a host
// program without the Odin sources (the command-line tool, the benchmark) can build the same
// inputs and consume the same output.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/rt_mi355x.h"

namespace {

struct d3 { double x, y, z; };
inline d3 from(rt_v3 v) { return {v.x, v.y, v.z}; }
inline rt_v3 to(d3 v) { return {v.x, v.y, v.z}; }
inline d3 add(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline d3 sub(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline d3 scale(d3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline d3 unit(d3 a) { double m = std::sqrt(dot3(a, a)); return {a.x / m, a.y / m, a.z / m}; }
inline d3 cross3(d3 a, d3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

// splitmix64: the generator behind the procedural scenes
struct SplitMix {
    uint64_t s;
    explicit SplitMix(uint64_t seed) : s(seed) {}
    uint64_t next_u64() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    double uniform() { return static_cast<double>(next_u64() >> 11) * (1.0 / 9007199254740992.0); } // [0,1)
    double range(double lo, double hi) { return (hi - lo) * uniform() + lo; }
};

struct SceneWriter {
    rt_sphere *spheres; rt_material *materials;
    int64_t cap_s, cap_m, ns = 0, nm = 0;
    int64_t material(int64_t type, double r, double g, double b, double param1) {
        if (nm < cap_m && materials) materials[nm] = {type, {r, g, b}, param1};
        return nm++;
    }
    void sphere(double x, double y, double z, double radius, int64_t material_index) {
        if (ns < cap_s && spheres) spheres[ns] = {{x, y, z}, radius, material_index};
        ++ns;
    }
    int32_t finish(int64_t *n_spheres, int64_t *n_materials) const {
        const bool fits = ns <= cap_s && nm <= cap_m && (spheres || ns == 0) && (materials || nm == 0);
        *n_spheres = ns; *n_materials = nm;
        return fits ? RT_OK : RT_ERR_BUFFER_TOO_SMALL;
    }
};

void camera(rt_camera_settings *c, d3 position, d3 target, int64_t image_width, double focus, double vfov_deg, double dof_deg, int64_t spp, int64_t bounces) {
    rt_v3 right, up, forward;
    rt_lookat(to(position), to(target), {0, 1, 0}, &right, &up, &forward);
    c->position = to(position); c->right = right; c->up = up; c->forward = forward;
    c->aspect_ratio = 16.0 / 9.0;
    c->image_size_x = static_cast<double>(image_width);
    c->image_size_y = c->image_size_x / c->aspect_ratio;
    c->focus_distance = focus;
    c->vfov = vfov_deg / 360.0;
    c->depth_of_field_angle = dof_deg / 360.0;
    c->samples_per_pixel = spp;
    c->max_ray_bounces = bounces;
}

bool bad_args(const rt_camera_settings *settings, const int64_t *n_spheres, const int64_t *n_materials) {
    return settings == nullptr || n_spheres == nullptr || n_materials == nullptr;
}

} // namespace

extern "C" {

// lookat  main.odin:65-71
void rt_lookat(rt_v3 position, rt_v3 target, rt_v3 axis_up, rt_v3 *right, rt_v3 *up, rt_v3 *forward) {
    d3 f = unit(sub(from(target), from(position)));
    d3 r = unit(cross3(f, from(axis_up)));
    d3 u = cross3(r, f);
    if (right) *right = to(r);
    if (up) *up = to(u);
    if (forward) *forward = to(f);
}

// calculate_camera_render_data  main.odin:450-485
void rt_calculate_camera_render_data(const rt_camera_settings *s, rt_camera_render_data *out) {
    if (s == nullptr || out == nullptr) return;
    const double tau = 6.283185307179586476925286766559;
    const double half_h = s->focus_distance * std::tan((s->vfov * 0.5) * tau);
    const double half_w = half_h * s->aspect_ratio;
    // pixel pitch on the focus plane; the vertical one is negative (image rows grow downwards)
    const double step_x = half_w * 2 / s->image_size_x * 1.0;
    const double step_y = half_h * 2 / s->image_size_y * -1.0;
    const d3 right = from(s->right), up = from(s->up), forward = from(s->forward), pos = from(s->position);
    out->position = s->position;
    out->pixel_delta_u = to(scale(right, step_x));
    out->pixel_delta_v = to(scale(up, step_y));
    const double first_x = half_w * -1.0 + step_x * 0.5;
    const double first_y = half_h * 1.0 + step_y * 0.5;
    out->view_plane_top_left_pixel_center = to(add(add(add(pos, scale(forward, s->focus_distance)), scale(right, first_x)), scale(up, first_y)));
    const double lens = s->focus_distance * std::tan((s->depth_of_field_angle * 0.5) * tau);
    out->dof_disk_u = to(scale(right, lens));
    out->dof_disk_v = to(scale(up, lens));
    out->depth_of_field_angle = s->depth_of_field_angle;
    out->samples_per_pixel = s->samples_per_pixel;
    out->max_ray_bounces = s->max_ray_bounces;
}

// build_dev_scene  main.odin:687-737
int32_t rt_build_dev_scene(int64_t image_width, rt_camera_settings *settings, rt_sphere *spheres, int64_t *n_spheres, rt_material *materials, int64_t *n_materials) {
    if (bad_args(settings, n_spheres, n_materials)) return RT_ERR_INVALID_ARGUMENT;
    SceneWriter w{spheres, materials, *n_spheres, *n_materials};
    const int64_t ground = w.material(RT_MATERIAL_LAMBERTIAN, 0.8, 0.8, 0.0, 0.0);
    const int64_t blue = w.material(RT_MATERIAL_LAMBERTIAN, 0.1, 0.2, 0.5, 0.0);
    const int64_t glass = w.material(RT_MATERIAL_DIELECTRIC, 0.0, 0.0, 0.0, 1.5);
    const int64_t bubble = w.material(RT_MATERIAL_DIELECTRIC, 0.0, 0.0, 0.0, 1.0 / 1.5);
    const int64_t gold = w.material(RT_MATERIAL_METALLIC, 0.8, 0.6, 0.2, 1.0);
    w.sphere(0.0, -100.5, -1.0, 100, ground);
    w.sphere(0.0, 0.0, -1.2, 0.5, blue);
    w.sphere(-1.0, 0.0, -1.0, 0.5, glass);
    w.sphere(-1.0, 0.0, -1.0, 0.4, bubble);
    w.sphere(1.0, 0.0, -1.0, 0.5, gold);
    camera(settings, {-2, 2, 1}, {0, 0, -1}, image_width > 0 ? image_width : 200, 3.4, 20.0, 10.0, 100, 50);
    return w.finish(n_spheres, n_materials);
}

int32_t rt_build_two_sphere_scene(int64_t image_width, rt_camera_settings *settings, rt_sphere *spheres, int64_t *n_spheres, rt_material *materials, int64_t *n_materials) {
    if (bad_args(settings, n_spheres, n_materials)) return RT_ERR_INVALID_ARGUMENT;
    SceneWriter w{spheres, materials, *n_spheres, *n_materials};
    const int64_t ground = w.material(RT_MATERIAL_LAMBERTIAN, 0.8, 0.8, 0.0, 0.0);
    const int64_t centre = w.material(RT_MATERIAL_LAMBERTIAN, 0.1, 0.2, 0.5, 0.0);
    w.sphere(0.0, -100.5, -1.0, 100, ground);
    w.sphere(0.0, 0.0, -1.0, 0.5, centre);
    camera(settings, {0, 0, 0}, {0, 0, -1}, image_width > 0 ? image_width : 400, 1.0, 90.0, 0.0, 100, 50);
    return w.finish(n_spheres, n_materials);
}

// build_final_scene  main.odin:739-789, draws from splitmix64(seed) in the reference's order
int32_t rt_build_final_scene(uint64_t seed, int64_t image_width, rt_camera_settings *settings, rt_sphere *spheres, int64_t *n_spheres, rt_material *materials, int64_t *n_materials) {
    if (bad_args(settings, n_spheres, n_materials)) return RT_ERR_INVALID_ARGUMENT;
    SceneWriter w{spheres, materials, *n_spheres, *n_materials};
    SplitMix rng(seed);
    w.material(RT_MATERIAL_LAMBERTIAN, 0.5, 0.5, 0.5, 0.0);
    w.material(RT_MATERIAL_DIELECTRIC, 0.0, 0.0, 0.0, 1.5);
    w.material(RT_MATERIAL_LAMBERTIAN, 0.4, 0.2, 0.1, 0.0);
    w.material(RT_MATERIAL_METALLIC, 0.7, 0.6, 0.5, 0.0);
    w.sphere(0.0, -1000, 0, 1000, 0);
    w.sphere(0, 1, 0, 1.0, 1);
    w.sphere(-4, 1, 0, 1.0, 2);
    w.sphere(4, 1, 0, 1.0, 3);
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const double cx = a + 0.9 * rng.uniform();
            const double cz = b + 0.9 * rng.uniform();
            const d3 off = sub({cx, 0.2, cz}, {4, 0.2, 0});
            if (!(std::sqrt(dot3(off, off)) > 0.9)) continue;
            const double choose = rng.uniform();
            int64_t m;
            if (choose < 0.8) { // diffuse: albedo = random_v3() * random_v3()
                const double p0 = rng.uniform(), p1 = rng.uniform(), p2 = rng.uniform();
                const double q0 = rng.uniform(), q1 = rng.uniform(), q2 = rng.uniform();
                m = w.material(RT_MATERIAL_LAMBERTIAN, p0 * q0, p1 * q1, p2 * q2, 0.0);
            } else if (choose < 0.95) { // metal: albedo in [0.5, 1)^3, fuzz in [0, 0.5)
                const double r = rng.range(0.5, 1), g = rng.range(0.5, 1), bl = rng.range(0.5, 1);
                const double fuzz = rng.range(0, 0.5);
                m = w.material(RT_MATERIAL_METALLIC, r, g, bl, fuzz);
            } else {
                m = w.material(RT_MATERIAL_DIELECTRIC, 0.0, 0.0, 0.0, 1.5);
            }
            w.sphere(cx, 0.2, cz, 0.2, m);
        }
    }
    camera(settings, {13, 2, 3}, {0, 0, 0}, image_width > 0 ? image_width : 1200, 10, 20.0, 0.6, 500, 50);
    return w.finish(n_spheres, n_materials);
}

int32_t rt_build_stress_scene(uint64_t seed, int64_t total_spheres, int64_t image_width, rt_camera_settings *settings, rt_sphere *spheres, int64_t *n_spheres, rt_material *materials, int64_t *n_materials) {
    if (bad_args(settings, n_spheres, n_materials) || total_spheres < 1) return RT_ERR_INVALID_ARGUMENT;
    SceneWriter w{spheres, materials, *n_spheres, *n_materials};
    SplitMix rng(seed);
    w.material(RT_MATERIAL_LAMBERTIAN, 0.5, 0.5, 0.5, 0.0);
    w.sphere(0.0, -1000, 0, 1000, 0);
    for (int64_t i = 1; i < total_spheres; ++i) {
        const double radius = rng.range(0.05, 0.2);
        const double cx = rng.range(-30, 30), cz = rng.range(-30, 30);
        const double choose = rng.uniform();
        int64_t m;
        if (choose < 0.8) {
            const double p0 = rng.uniform(), p1 = rng.uniform(), p2 = rng.uniform();
            const double q0 = rng.uniform(), q1 = rng.uniform(), q2 = rng.uniform();
            m = w.material(RT_MATERIAL_LAMBERTIAN, p0 * q0, p1 * q1, p2 * q2, 0.0);
        } else if (choose < 0.95) {
            const double r = rng.range(0.5, 1), g = rng.range(0.5, 1), bl = rng.range(0.5, 1);
            m = w.material(RT_MATERIAL_METALLIC, r, g, bl, rng.range(0, 0.5));
        } else {
            m = w.material(RT_MATERIAL_DIELECTRIC, 0.0, 0.0, 0.0, 1.5);
        }
        w.sphere(cx, radius, cz, radius, m);
    }
    camera(settings, {13, 2, 3}, {0, 0, 0}, image_width > 0 ? image_width : 1920, 10, 20.0, 0.6, 256, 50);
    return w.finish(n_spheres, n_materials);
}

// the quantiser of serialize_ppm  main.odin:812-814
static inline uint8_t quantize(double x) { return static_cast<uint8_t>(static_cast<int64_t>(256 * std::min(x, 0.999))); }

int32_t rt_quantize_rgb8(const rt_v3 *image_data, int64_t n_pixels, uint8_t *rgb8) {
    if ((image_data == nullptr || rgb8 == nullptr) && n_pixels > 0) return RT_ERR_INVALID_ARGUMENT;
    for (int64_t i = 0; i < n_pixels; ++i) {
        rgb8[3 * i + 0] = quantize(image_data[i].x);
        rgb8[3 * i + 1] = quantize(image_data[i].y);
        rgb8[3 * i + 2] = quantize(image_data[i].z);
    }
    return RT_OK;
}

// serialize_ppm  main.odin:791-824
int64_t rt_serialize_ppm(const rt_v3 *image_data, int64_t width, int64_t height, char *dst, int64_t capacity) {
    if (width < 0 || height < 0) return RT_ERR_INVALID_ARGUMENT;
    // the image size limit of the render entry points, applied before any size is computed from the arguments
    if (width > (int64_t(1) << 30) || height > (int64_t(1) << 30) || width * height > (int64_t(1) << 30)) return RT_ERR_UNSUPPORTED;
    char header[64];
    const int header_len = std::snprintf(header, sizeof header, "P3\n%lld %lld\n255\n", (long long)width, (long long)height);
    const int64_t need = header_len + width * height * 12;
    if (dst == nullptr || capacity < need || (image_data == nullptr && width * height > 0)) return need;
    std::memcpy(dst, header, header_len);
    char *out = dst + header_len;
    // every value is three columns, right aligned, followed by ' ' or, after blue, '\n'
    static const char digits[] = "0123456789";
    for (int64_t i = 0; i < width * height; ++i) {
        const uint8_t rgb[3] = {quantize(image_data[i].x), quantize(image_data[i].y), quantize(image_data[i].z)};
        for (int c = 0; c < 3; ++c) {
            const unsigned v = rgb[c];
            out[0] = v >= 100 ? digits[v / 100] : ' ';
            out[1] = v >= 10 ? digits[(v / 10) % 10] : ' ';
            out[2] = digits[v % 10];
            out[3] = c == 2 ? '\n' : ' ';
            out += 4;
        }
    }
    return need;
}

} // extern "C"
