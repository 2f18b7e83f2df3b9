// Command-line host for the GPU renderer: what `main` of the reference does (main.odin:826-852)
// through the C ABI -- build a scene, compute the camera, render, print a P3 PPM to stdout --
// with the scene, size, sample count and seed selectable at run time.
//   rt_cli [--scene dev|two|final|stress] [--width W] [--spp N] [--depth D] [--seed S]
//          [--gpus G] [--precision mixed|mixed-fixed|f32|f64|f32-literal] [--stochastic] [--stats] [--host-ppm] [--rounds N] > out.ppm
// --rounds N is bench_materials of the reference (main_test.odin:8-46, :72-81): N renders of the scene through
// the one-shot entry point, the last image to stdout, and the reference's statistics line to stderr (its
// "bytes" are pixels, main_test.odin:24).
// The PPM text is formatted on the GPU (rt_render_ppm); --host-ppm takes the reference's route
// instead: the f64 image comes back and rt_serialize_ppm formats it on the host.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_mi355x.h"

int main(int argc, char **argv) {
    std::string scene = "dev", precision = "mixed";
    int64_t width = 0, spp = 0, depth = -1;
    uint64_t seed = 1;
    int gpus = 1;
    long long rounds = 0;
    bool stochastic = false, want_stats = false, host_ppm = false;
    for (int i = 1; i < argc; ++i) {
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--scene")) scene = next();
        else if (!std::strcmp(argv[i], "--width")) width = std::atoll(next());
        else if (!std::strcmp(argv[i], "--spp")) spp = std::atoll(next());
        else if (!std::strcmp(argv[i], "--depth")) depth = std::atoll(next());
        else if (!std::strcmp(argv[i], "--seed")) seed = std::strtoull(next(), nullptr, 0);
        else if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(next());
        else if (!std::strcmp(argv[i], "--precision")) precision = next();
        else if (!std::strcmp(argv[i], "--stochastic")) stochastic = true;
        else if (!std::strcmp(argv[i], "--stats")) want_stats = true;
        else if (!std::strcmp(argv[i], "--host-ppm")) host_ppm = true;
        else if (!std::strcmp(argv[i], "--rounds")) rounds = std::atoll(next());
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    rt_camera_settings settings;
    std::vector<rt_sphere> spheres(16384);
    std::vector<rt_material> materials(16384);
    int64_t ns = static_cast<int64_t>(spheres.size()), nm = static_cast<int64_t>(materials.size());
    int32_t rc;
    if (scene == "dev") rc = rt_build_dev_scene(width, &settings, spheres.data(), &ns, materials.data(), &nm);
    else if (scene == "two") rc = rt_build_two_sphere_scene(width, &settings, spheres.data(), &ns, materials.data(), &nm);
    else if (scene == "final") rc = rt_build_final_scene(seed, width, &settings, spheres.data(), &ns, materials.data(), &nm);
    else if (scene == "stress") rc = rt_build_stress_scene(5, 10000, width, &settings, spheres.data(), &ns, materials.data(), &nm);
    else { std::fprintf(stderr, "unknown scene %s\n", scene.c_str()); return 2; }
    if (rc != RT_OK) { std::fprintf(stderr, "scene builder failed: %d\n", rc); return 1; }
    if (spp > 0) settings.samples_per_pixel = spp;
    if (depth >= 0) settings.max_ray_bounces = depth;

    const int64_t w = static_cast<int64_t>(settings.image_size_x), h = static_cast<int64_t>(settings.image_size_y);
    rt_camera_render_data cam;
    rt_calculate_camera_render_data(&settings, &cam);

    rt_options opts;
    rt_options_default(&opts);
    opts.seed = seed;
    opts.rng_mode = stochastic ? RT_RNG_STOCHASTIC : RT_RNG_DETERMINISTIC;
    if (precision == "f64") opts.precision = RT_PRECISION_F64;
    else if (precision == "mixed") opts.precision = RT_PRECISION_MIXED; // the library's default: the reference's f64 image
    else if (precision == "mixed-fixed") opts.precision = RT_PRECISION_MIXED_FIXED;
    else if (precision == "f32-literal") opts.precision = RT_PRECISION_F32_LITERAL;
    else if (precision == "f32") opts.precision = RT_PRECISION_F32;
    else { std::fprintf(stderr, "unknown precision %s\n", precision.c_str()); return 2; }
    opts.device_count = gpus;
    uint64_t stats[RT_STATS_COUNT] = {0};
    opts.stats = stats;
    // main renders with print_progress = true (main.odin:841), bench_materials with false (main_test.odin:28)
    if (rounds <= 0) opts.flags |= RT_FLAG_PRINT_PROGRESS;
    std::fprintf(stderr, "%s: %s scene, %lld spheres, %lldx%lld, %lld spp, depth %lld, %d GPU(s)\n", rt_version(), scene.c_str(), (long long)ns,
                 (long long)w, (long long)h, (long long)settings.samples_per_pixel, (long long)settings.max_ray_bounces, gpus);
    if (rounds > 0) {
        std::vector<rt_v3> image(static_cast<size_t>(w * h));
        const auto b0 = std::chrono::steady_clock::now();
        double ns_first = 0.0;
        for (long long round = 1; round <= rounds; ++round) {
            std::fprintf(stderr, "\rRound %lld/%lld", round, rounds);
            rc = rt_render_tiled(image.data(), w, h, &cam, spheres.data(), ns, materials.data(), nm, &opts);
            if (rc != RT_OK) { std::fprintf(stderr, "\nrt_render_tiled failed (%d): %s\n", rc, rt_last_error()); return 1; }
            if (round == 1) ns_first = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - b0).count();
        }
        const double ns_total = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - b0).count();
        std::fprintf(stderr, "\rDone.        \n");
        // the first round loads the GPU code and uploads the world; the library keeps both for the rounds that follow
        if (rounds > 1) std::fprintf(stderr, "first round %.2f ms, later rounds %.3f ms each (%.1f rounds/s)\n", ns_first * 1e-6, (ns_total - ns_first) * 1e-6 / double(rounds - 1),
                                     double(rounds - 1) / ((ns_total - ns_first) * 1e-9));
        const int64_t size = rt_serialize_ppm(image.data(), w, h, nullptr, 0);
        std::vector<char> text(static_cast<size_t>(size));
        rt_serialize_ppm(image.data(), w, h, text.data(), size);
        std::fwrite(text.data(), 1, text.size(), stdout);
        const double processed = double(rounds) * double(w * h); // options.bytes = len(image.data): pixels
        // benchmark_print (main_test.odin:72-81): "[%v] %v rounds, %v bytes processed in %v ns, %5.3f rounds/s, %5.3f MiB/s" with the
        // benchmark's name, "materials" (main_test.odin:59), for the scene it renders
        std::fprintf(stderr, "[%s] %lld rounds, %.0f bytes processed in %.0f ns, %5.3f rounds/s, %5.3f MiB/s\n", scene == "dev" ? "materials" : scene.c_str(), rounds,
                     processed, ns_total, double(rounds) / (ns_total * 1e-9), processed / (ns_total * 1e-9) / (1024.0 * 1024.0));
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t need = rt_serialize_ppm(nullptr, w, h, nullptr, 0);
    std::vector<char> ppm(static_cast<size_t>(need));
    if (host_ppm) {
        std::vector<rt_v3> image(static_cast<size_t>(w * h));
        rc = rt_render_tiled(image.data(), w, h, &cam, spheres.data(), ns, materials.data(), nm, &opts);
        if (rc != RT_OK) { std::fprintf(stderr, "rt_render_tiled failed (%d): %s\n", rc, rt_last_error()); return 1; }
        rt_serialize_ppm(image.data(), w, h, ppm.data(), need);
    } else {
        const int64_t got = rt_render_ppm(w, h, &cam, spheres.data(), ns, materials.data(), nm, &opts, ppm.data(), need);
        if (got != need) { std::fprintf(stderr, "rt_render_ppm failed (%lld): %s\n", (long long)got, rt_last_error()); return 1; }
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (want_stats)
        std::fprintf(stderr, "%.3f s wall to the serialised image (%.1f Msamples/s incl. set-up, copies and PPM text), kernel %.3f ms, %.3f segments/sample, %llu sphere tests\n", seconds,
                     w * h * settings.samples_per_pixel / seconds / 1e6, stats[RT_STAT_KERNEL_NS] / 1e6,
                     double(stats[RT_STAT_SEGMENTS]) / double(stats[RT_STAT_SAMPLES]), (unsigned long long)stats[RT_STAT_SPHERE_TESTS]);
    std::fwrite(ppm.data(), 1, ppm.size(), stdout);
    return 0;
}
