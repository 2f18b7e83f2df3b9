/*
 * rt_mi355x.h -- C ABI of the MI355X (gfx950) path-tracing library.
 *
 * The reference program (ray-tracer/main.odin) has no FFI of its own: its hot
 * path is the Odin procedure `render_region` (main.odin:510-591) that `main`
 * calls at main.odin:840-844 and the benchmark calls at main_test.odin:27-31.
 * This header declares the C entry points an Odin `foreign import` block binds
 * in place of those calls.  Every struct below has the field order, field
 * types and therefore the byte layout of the Odin struct it names, so the Odin
 * host can pass `raw_data(world.spheres)` etc. without conversion.
 *
 * All pointers are owned by the caller.  Nothing is retained after a call
 * returns except what an explicit handle (rt_world) owns.  No function throws
 * across the boundary: every int32_t return is RT_OK or a negative RT_ERR_*
 * and rt_last_error() returns a thread-local message for the last failure.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------- */
#define RT_OK 0
#define RT_ERR_INVALID_ARGUMENT (-1) /* null pointer, bad size, bad rect, bad option value      */
#define RT_ERR_INVALID_SCENE (-2)    /* material_index out of range, fuzz outside [0,1], r == 0 */
#define RT_ERR_DEVICE (-3)           /* a HIP call failed (no GPU, out of memory, launch error) */
#define RT_ERR_UNSUPPORTED (-4)      /* a documented limit was exceeded                          */
#define RT_ERR_BUFFER_TOO_SMALL (-5) /* caller-provided capacity is insufficient                 */

/* ---- data that crosses the boundary (layouts == the Odin structs) ------ */

/* v3 :: [3]f64                                             main.odin:19   */
typedef struct { double x, y, z; } rt_v3;

/* sphere :: struct {center: v3, radius: f64, material_index: i64}   main.odin:487-491 (40 B) */
typedef struct { rt_v3 center; double radius; int64_t material_index; } rt_sphere;

/* material_type :: enum i64 {lambertian, metallic, dielectric}      main.odin:268-272 */
#define RT_MATERIAL_LAMBERTIAN 0
#define RT_MATERIAL_METALLIC 1
#define RT_MATERIAL_DIELECTRIC 2
/* metallic_proc as compiled with -define:USE_METAL_FRESNEL=true (main.odin:324-347) and the
 * METAL_FRESNEL_KIND the reference selects (2, :332): the attenuation becomes
 * lerp(albedo, 1, reflectance_fresnel(cos, sin, 1.0, 1.27035)) (main.odin:84-109, :341-347).
 * The reference switches all metals at compile time; here it is a material type, so a host built
 * with that flag passes 3 where it passes 1 today.  Same data: albedo, param1 = fuzz. */
#define RT_MATERIAL_METALLIC_FRESNEL 3
/* The other two METAL_FRESNEL_KINDs of that block, which the reference does not select (main.odin:332):
 * kind 0, reflectance_schlick_approximation(cos, r0) with r0 from 1.0/1.27035 (main.odin:334-335, :111-122), and
 * kind 1, reflectance_hoffman_approximation(cos, r0, 0.5) (main.odin:337-339, :132-144).
 * RT_MATERIAL_METALLIC_HOFFMAN is kind 1 AS THE REFERENCE'S TEXT COMPUTES IT, so that a host built with
 * METAL_FRESNEL_KIND = 1 gets the image of its own CPU path: the Schlick term is called with its arguments swapped,
 * reflectance_schlick_approximation(r0, cos_i) (main.odin:133), i.e. cos + (1 - cos)(1 - r0)^5, and the factors
 * 823543 / 46656 and 49 / 6 are untyped integer divisions, 17 and 8 (main.odin:142).
 * RT_MATERIAL_METALLIC_HOFFMAN_PAPER is the approximation as the cited papers define it: Schlick's arguments in their
 * declared order and the factors 7^7/6^6 and 49/6 as real quotients (with h = 0.5 the reflectance at cos = 1/7 is then
 * exactly 0.5); no counterpart in the reference.  In both, math.pow(x, 6) -- core:math, not in the reference's
 * checkout -- is taken as ((x*x)*x)^2.  No fixture of the reference covers kinds 0 and 1: parity unpinned. */
#define RT_MATERIAL_METALLIC_SCHLICK 4
#define RT_MATERIAL_METALLIC_HOFFMAN 5
#define RT_MATERIAL_METALLIC_HOFFMAN_PAPER 6

/* material :: struct {type: material_type, data: {albedo: v3, param1: f64}}
 * param1 = fuzz for metallic, refractive index for dielectric       main.odin:274-289 (40 B) */
typedef struct { int64_t type; rt_v3 albedo; double param1; } rt_material;

/* camera_render_data                                        main.odin:440-448 (168 B)
 * depth_of_field_angle is in turns (1 turn = 2*pi rad, main.odin:21-23). */
typedef struct {
    rt_v3 position;
    rt_v3 pixel_delta_u, pixel_delta_v;
    rt_v3 view_plane_top_left_pixel_center;
    rt_v3 dof_disk_u, dof_disk_v;
    double depth_of_field_angle;
    int64_t samples_per_pixel;
    int64_t max_ray_bounces;
} rt_camera_render_data;

/* camera_settings                                           main.odin:425-438 (160 B)
 * image_size is v2 ([2]f64); vfov and depth_of_field_angle are in turns. */
typedef struct {
    rt_v3 position, right, up, forward;
    double aspect_ratio;
    double image_size_x, image_size_y;
    double focus_distance;
    double vfov;
    double depth_of_field_angle;
    int64_t samples_per_pixel;
    int64_t max_ray_bounces;
} rt_camera_settings;

/* rect: half-open pixel rectangle                            main.odin:503-508 */
typedef struct { int64_t min_x, min_y, one_past_max_x, one_past_max_y; } rt_rect;

/* ---- options ------------------------------------------------------------ */

/* arithmetic the kernel computes in */
#define RT_PRECISION_F32 0         /* f32 paths, fused multiply-adds in the ray/sphere test (a conservative filter in
                                      seven operations per sphere, then the roots of the spheres it keeps), exact Q32.32
                                      sample sums, f64 mean and square root (throughput path).  The sphere a scattered
                                      segment starts on is tested with c = 0 (main.odin:247 evaluates to ~1e-15
                                      there in f64; its f32 rounding would give spurious self-intersections)   */
#define RT_PRECISION_F64 1         /* f64, the reference's literal operation order (main.odin:19 is f64)      */
#define RT_PRECISION_F32_LITERAL 2 /* f32, the reference's literal operation order, no fused multiply-adds    */
#define RT_PRECISION_MIXED 3       /* the image of RT_PRECISION_F64 (with one sample chunk: f64, literal order,
                                      colours summed in sample order, main.odin:575) bit for bit, traced faster:
                                      the f32 filter scan of RT_PRECISION_F32 runs on the f64 ray rounded to f32
                                      -- a sphere is skipped only when its discriminant
                                      (main.odin:249) is negative by more than the f32 rounding can explain --
                                      and every sphere it keeps goes through ray_sphere_intersection as written
                                      (main.odin:245-261) in f64; hit record, scattering and colour are f64.
                                      Worlds outside 2^-20 <= |r|, |c| + |r| <= 2^20 are traced by the
                                      RT_PRECISION_F64 kernel instead (same image).  RT_FETCH_AUTO / _SCALAR.
                                      THE DEFAULT (rt_options_default, opts == NULL).                             */
#define RT_PRECISION_MIXED_FIXED 4 /* the paths of RT_PRECISION_MIXED (f64, bit for bit the reference's), with
                                      `pixel_color += sample_color` (main.odin:575) carried out in fixed point: each
                                      f64 sample colour is truncated to a multiple of 2^-64 and added as an integer
                                      (two 64-bit words per channel, atomic adds), so the sum is exact, does not
                                      depend on the order in which lanes, waves, calls or GPUs finish samples, and
                                      needs no buffer beyond the accumulators; the mean is that sum rounded to f64
                                      ONCE, times 1/spp.  It differs from the reference's sequential f64 sum by that
                                      sum's own rounding (a few 2^-53 relative): not bitwise the RT_PRECISION_F64
                                      image, but within 1e-12 relative of it per linear pixel.                   */

/* how the closest hit (main.odin:536-546) is found */
#define RT_FETCH_AUTO 0   /* = RT_FETCH_SCALAR                                                     */
#define RT_FETCH_SCALAR 1 /* list scan, records through scalar-cache loads into SGPRs              */
#define RT_FETCH_LDS 2    /* list scan, SoA records staged once per workgroup into LDS, broadcast  */
#define RT_FETCH_BVH 3    /* bounding-volume hierarchy over the spheres instead of the list (the
                             TODO at main.odin:559); RT_PRECISION_F32 and RT_PRECISION_MIXED; the image is
                             the list scan's bit for bit (with RT_PRECISION_MIXED: the f64 image; also
                             RT_PRECISION_MIXED_FIXED),
                             RT_STAT_SPHERE_TESTS counts the tests carried out.
                             A hierarchy of at most 1280 nodes (about 640 spheres) is staged into LDS  */
#define RT_FETCH_BVH_GLOBAL 4 /* the same with the hierarchy read from global memory whatever its size
                             (what RT_FETCH_BVH does for larger worlds)                              */

#define RT_RNG_STOCHASTIC 0    /* seed is mixed with a per-process call counter and the clock */
#define RT_RNG_DETERMINISTIC 1 /* the counter RNG is keyed by `seed` alone: same inputs, same image */

/* rt_options.flags */
#define RT_FLAG_PRINT_PROGRESS 1 /* the $print_progress parameter of render_region (main.odin:510): the call prints the
                                    reference's closing "Done." line to stderr (main.odin:590).  The per-scanline
                                    "Scanlines remaining" counter (main.odin:514) has no counterpart: the GPU does not
                                    render scanline by scanline                                                      */

/* indices into the optional stats array */
#define RT_STAT_SEGMENTS 0     /* ray segments traced = executed iterations of the bounce loop main.odin:535 */
#define RT_STAT_SPHERE_TESTS 1 /* segments * n_spheres = calls of ray_sphere_intersection main.odin:541   */
#define RT_STAT_KERNEL_NS 2    /* device time of the render kernels of this call, from HIP events          */
#define RT_STAT_SAMPLES 3      /* camera rays started                                                       */
#define RT_STAT_WAVE_ITERATIONS 4 /* sum over wave64s of their segment-loop trip count: segments / (64 * this)
                                  is the share of lanes that had a live ray in the closest-hit scan         */
#define RT_STAT_BVH_NODES 5    /* RT_FETCH_BVH: bounding boxes tested, summed over lanes                    */
#define RT_STAT_BVH_TESTS 6    /* RT_FETCH_BVH: ray/sphere tests carried out, summed over lanes             */
#define RT_STATS_COUNT 8

typedef struct {
    uint64_t seed;
    int32_t rng_mode;      /* RT_RNG_*                                                         */
    int32_t precision;     /* RT_PRECISION_*                                                   */
    int32_t sphere_fetch;  /* RT_FETCH_*                                                       */
    int32_t device;        /* HIP device ordinal, or -1 for the calling thread's current one  */
    int32_t device_count;  /* rt_render_region only: spread 8-row strips over this many GPUs
                              starting at `device` (0 and 1 both mean one GPU).  With
                              RT_MI355X_WRAP_DEVICES=1 in the environment the ordinals wrap
                              around the visible GPUs (a testing aid for boxes with fewer)   */
    int32_t sample_chunks; /* split samples_per_pixel over this many waves per tile; 0 = auto */
    int32_t flags;         /* RT_FLAG_*                                                        */
    int32_t reserved;      /* 0                                                                */
    rt_v3 *linear_out;     /* optional, width*height: pre-gamma sample means of region pixels  */
    uint64_t *stats;       /* optional, RT_STATS_COUNT counters for this call                  */
} rt_options;

/* ---- library ------------------------------------------------------------ */

const char *rt_version(void);
/* message of the calling thread's last failing call; never NULL */
const char *rt_last_error(void);
/* number of HIP devices visible, or a negative RT_ERR_DEVICE */
int32_t rt_device_count(void);
void rt_options_default(rt_options *opts);

/* ---- the drop-in entry points -------------------------------------------- */

/*
 * Replaces render_region (main.odin:510-591; call sites main.odin:841,
 * main_test.odin:28 and, per tile, main.odin:609).
 *
 * For every pixel (u, v) of `region` traces cam->samples_per_pixel paths of
 * at most cam->max_ray_bounces + 1 segments through the sphere list and
 * writes the gamma-2 colour (sqrt of the sample mean, main.odin:577-586) to
 * image_data[u + v * width].  Pixels outside `region` are not touched.
 * Host pointers in, host pointers out; synchronous; safe to call from several
 * threads (calls are serialised per device).
 * opts may be NULL (RT_PRECISION_MIXED -- the image of the reference's f64 arithmetic --, deterministic,
 * seed 0, current device: what rt_options_default sets).
 */
int32_t rt_render_region(rt_v3 *image_data, int64_t width, int64_t height, rt_rect region,
                         const rt_camera_render_data *cam,
                         const rt_sphere *spheres, int64_t n_spheres,
                         const rt_material *materials, int64_t n_materials,
                         const rt_options *opts);

/*
 * Replaces render_tiled (main.odin:635-685; call sites main.odin:843,
 * main_test.odin:30): renders the whole image.  The reference cuts the image
 * into 2x2-pixel CPU tasks; here the GPU grid of 8x8-pixel wave tiles is the
 * tiling, so this is rt_render_region over {0, 0, width, height}.
 */
int32_t rt_render_tiled(rt_v3 *image_data, int64_t width, int64_t height,
                        const rt_camera_render_data *cam,
                        const rt_sphere *spheres, int64_t n_spheres,
                        const rt_material *materials, int64_t n_materials,
                        const rt_options *opts);

/* ---- device-resident form of the same path -------------------------------- */
/*
 * The reference keeps `world` (main.odin:493-496) alive across the 100
 * render_region calls of its benchmark (main_test.odin:13-32).  rt_world is
 * that object on the GPU: validated, converted from AoS f64 to SoA f32 + f64
 * and uploaded once.
 */
typedef struct rt_world_s rt_world;

int32_t rt_world_create(const rt_sphere *spheres, int64_t n_spheres,
                        const rt_material *materials, int64_t n_materials,
                        int32_t device, rt_world **out);
int32_t rt_world_destroy(rt_world *world);

/*
 * The loops of render_region up to and including `pixel_color += sample_color`
 * (main.odin:513-576) for a subset of the work, with everything in device
 * memory:
 *   pixels : the pixels of `region` whose 8-row strip s = (v - region.min_y) /
 *            strip_rows satisfies s % strip_count == strip_index
 *            (strip_count 1 = the whole region; strip_rows a multiple of 8);
 *   samples: the global sample indices [sample_begin, sample_begin+sample_count).
 * d_accum is a device buffer of width*height*3 elements indexed
 * [(u + v*width)*3 + channel]; the call overwrites the entries of its pixels
 * with the SUM of its samples' colours.  Element type by precision:
 *   RT_PRECISION_MIXED       double, summed in sample order whatever sample_chunks
 *                            says (the colours of up to RT_MI355X_STAGING_MIB -- default
 *                            4096 -- MiB worth of samples, 24 bytes each, are staged in a
 *                            buffer the world owns and added by a second kernel; more samples
 *                            pass through it in windows that overlap on two streams of the
 *                            world's own, which the caller's stream waits for; an allocation
 *                            the device refuses is retried at half the size);
 *   RT_PRECISION_MIXED_FIXED two uint64_t {a, b} per element (16 bytes): the exact sum of the
 *                            colours, each truncated to a multiple of 2^-64, as a * 2^-32 +
 *                            b * 2^-64.  Only that value is defined, not the split between the
 *                            words; element-wise integer addition of two such buffers (a split
 *                            of the samples over calls or GPUs) adds the values;
 *   RT_PRECISION_F32         uint64_t, Q32.32 fixed point (each f32 sample colour is
 *                            truncated to 2^-32 and added exactly, so sums over any
 *                            split of the samples -- chunks, calls, GPUs -- are equal);
 *   RT_PRECISION_F64         double, summed in sample order as main.odin:575;
 *   RT_PRECISION_F32_LITERAL float, summed in sample order.
 * cam->samples_per_pixel is ignored here.  d_stats is NULL or a device array
 * of RT_STATS_COUNT u64 the kernel adds RT_STAT_SEGMENTS / RT_STAT_SAMPLES
 * into.  `stream` is a hipStream_t (NULL = the default stream); the call is
 * asynchronous with respect to the host.  Because the RNG is keyed by the
 * global pixel index and the global sample index, any partition of pixels or
 * samples over calls, streams or GPUs traces exactly the same paths.
 * Calls on one world may be in flight on several streams at once: every call
 * has its own work-queue head and timing events (a ring of 8; the ninth call
 * first waits on the host for the first to end), and calls that need the
 * world's scratch buffer (RT_PRECISION_F64 / _F32_LITERAL with more than one
 * sample chunk, RT_PRECISION_MIXED) are ordered one after the other on the
 * device.  Two calls that write the same d_accum entries race, as any two
 * writers would.
 */
int32_t rt_render_accumulate(rt_world *world, const rt_camera_render_data *cam,
                             int64_t width, int64_t height, rt_rect region,
                             int64_t strip_rows, int32_t strip_index, int32_t strip_count,
                             int64_t sample_begin, int64_t sample_count,
                             uint64_t seed, int32_t precision, int32_t sphere_fetch,
                             int32_t sample_chunks,
                             void *d_accum, uint64_t *d_stats, void *stream);

/*
 * The tail of the pixel loop (main.odin:577-586) on the device: for the
 * pixels selected as above, mean = accum * (1 / total_samples), and
 * d_image[u + v*width] = sqrt(mean) per channel (device rt_v3 array);
 * d_linear, if not NULL, receives the mean itself.  d_accum has the element
 * type rt_render_accumulate documents for `precision`.
 */
int32_t rt_finalize(rt_world *world, const void *d_accum, int32_t precision,
                    int64_t width, int64_t height, rt_rect region,
                    int64_t strip_rows, int32_t strip_index, int32_t strip_count,
                    int64_t total_samples, rt_v3 *d_image, rt_v3 *d_linear, void *stream);

/*
 * The same tail followed by the body of serialize_ppm (main.odin:806-820) on
 * the device, for the whole image: for the pixels of the selected strips,
 * d_text + 12 * (u + v*width) receives the 12 bytes "RRR GGG BBB\n", each
 * channel u8(256 * min(sqrt(mean), 0.999)) (main.odin:812-814) right-aligned
 * in three columns.  d_text is a 4-byte-aligned device buffer of
 * width*height*12 bytes; with the header "P3\n{w} {h}\n255\n" in front it is
 * the file serialize_ppm produces from the image rt_finalize writes.
 */
int32_t rt_finalize_ppm(rt_world *world, const void *d_accum, int32_t precision,
                        int64_t width, int64_t height,
                        int64_t strip_rows, int32_t strip_index, int32_t strip_count,
                        int64_t total_samples, char *d_text, void *stream);

/* device time in ns of the render kernels launched for `world` by the last
 * rt_render_accumulate on it (blocks until they finish); <0 on error */
int64_t rt_world_last_kernel_ns(rt_world *world);

/*
 * rt_render_accumulate for callers that cannot allocate device memory (code
 * under HIP graph capture, serving loops, harnesses that forbid hipMalloc).
 * No counterpart in the reference.  Same arguments, same result in d_accum,
 * but the scratch the kernels need -- the work-queue head, the partial sums
 * of sample chunks 1.. (RT_PRECISION_F64 / _F32_LITERAL) or the staged sample
 * colours (RT_PRECISION_MIXED) -- lives in one fixed 32 MiB array per GPU
 * that the library holds from load until the process ends, instead of in
 * buffers the world allocates and grows.  The call allocates and frees
 * nothing.  Limits: (sample_chunks - 1) * width * height * 3 elements of the
 * precision's accumulator type, or in mixed mode one sample of the call's
 * pixels (owned 8x8 tiles * 64 * 24 bytes), must fit rt_static_scratch_bytes();
 * RT_ERR_UNSUPPORTED otherwise (pass a smaller sample_chunks -- 1 needs no
 * slabs -- or fewer pixels per call), and for RT_FETCH_BVH unless an earlier
 * rt_render_accumulate on this world has built the hierarchy.
 * RT_PRECISION_F32 and RT_PRECISION_MIXED_FIXED need only the queue head and are
 * never refused for size.
 * Every call on a GPU shares that GPU's array: calls that overlap in time on
 * one GPU race on it (whatever world or stream they use); calls issued one
 * after the other on one stream do not.  While `stream` is being captured into
 * a HIP graph the call records no events and does not synchronise with the
 * host; rt_world_last_kernel_ns then keeps referring to the last call made
 * outside a capture.
 */
int32_t rt_render_accumulate_static(rt_world *world, const rt_camera_render_data *cam,
                                    int64_t width, int64_t height, rt_rect region,
                                    int64_t strip_rows, int32_t strip_index, int32_t strip_count,
                                    int64_t sample_begin, int64_t sample_count,
                                    uint64_t seed, int32_t precision, int32_t sphere_fetch,
                                    int32_t sample_chunks,
                                    void *d_accum, uint64_t *d_stats, void *stream);
/* bytes of that array available for slabs / staged colours */
int64_t rt_static_scratch_bytes(void);

/* ---- host-side helpers the callers of the path use ------------------------ */

/* lookat (main.odin:65-71) */
void rt_lookat(rt_v3 position, rt_v3 target, rt_v3 axis_up, rt_v3 *right, rt_v3 *up, rt_v3 *forward);
/* calculate_camera_render_data (main.odin:450-485) */
void rt_calculate_camera_render_data(const rt_camera_settings *settings, rt_camera_render_data *out);

/*
 * Scene builders.  spheres/materials are caller arrays of *n_spheres /
 * *n_materials capacity; on return the counts hold the number written
 * (RT_ERR_BUFFER_TOO_SMALL leaves the required counts there).
 *   dev   : build_dev_scene (main.odin:687-737), verbatim data.
 *   two   : ground + centre lambertian spheres, camera at the origin, vfov 90 deg.
 *   final : the generation rule of build_final_scene (main.odin:739-789).  The
 *           reference draws from Odin's global generator after rand.reset(1);
 *           that stream is not reproducible outside Odin, so the draws come
 *           from splitmix64(seed) in the same order.
 *   stress: ground + (n-1) small spheres on a 60x60 field, same material mix.
 * image_width sets image_size (height = width / aspect_ratio, as main.odin:729).
 */
int32_t rt_build_dev_scene(int64_t image_width, rt_camera_settings *settings,
                           rt_sphere *spheres, int64_t *n_spheres,
                           rt_material *materials, int64_t *n_materials);
int32_t rt_build_two_sphere_scene(int64_t image_width, rt_camera_settings *settings,
                                  rt_sphere *spheres, int64_t *n_spheres,
                                  rt_material *materials, int64_t *n_materials);
int32_t rt_build_final_scene(uint64_t seed, int64_t image_width, rt_camera_settings *settings,
                             rt_sphere *spheres, int64_t *n_spheres,
                             rt_material *materials, int64_t *n_materials);
int32_t rt_build_stress_scene(uint64_t seed, int64_t total_spheres, int64_t image_width,
                              rt_camera_settings *settings,
                              rt_sphere *spheres, int64_t *n_spheres,
                              rt_material *materials, int64_t *n_materials);

/*
 * serialize_ppm (main.odin:791-824): "P3\n{w} {h}\n255\n" then 12 bytes per
 * pixel, each channel u8(256 * min(x, 0.999)) right-aligned in 3 columns.
 * Returns the number of bytes the image needs; writes them if capacity allows
 * (dst may be NULL to query).  No terminating NUL is written.  Images of more
 * than 2^30 pixels: RT_ERR_UNSUPPORTED (here and in rt_render_ppm).
 */
int64_t rt_serialize_ppm(const rt_v3 *image_data, int64_t width, int64_t height,
                         char *dst, int64_t capacity);
/*
 * main (main.odin:840-848) from the render call to the serialised image in
 * one step: rt_render_tiled followed by serialize_ppm, with the quantiser and
 * the text formatting done on the GPU, so that 12 bytes per pixel cross PCIe
 * instead of 24 and the host formats nothing.  Byte for byte the output of
 * rt_serialize_ppm on the image rt_render_tiled returns.  Same size/capacity
 * protocol as rt_serialize_ppm; a negative RT_ERR_* on failure.
 */
int64_t rt_render_ppm(int64_t width, int64_t height, const rt_camera_render_data *cam,
                      const rt_sphere *spheres, int64_t n_spheres,
                      const rt_material *materials, int64_t n_materials,
                      const rt_options *opts, char *dst, int64_t capacity);
/*
 * rt_render_region, rt_render_tiled and rt_render_ppm keep, per GPU, what a call needed -- the
 * uploaded world (reused while the caller passes the same spheres and materials, byte for byte), the
 * accumulator, image and text buffers, a stream -- so that calling them in a loop, as the reference's
 * benchmark does 100 times (main_test.odin:25-32), costs little more than the kernels.  This frees it
 * all (every GPU); the next call sets up again.  No counterpart in the reference.
 */
int32_t rt_release_cache(void);
/* the quantiser of main.odin:812-814 alone: rgb8 gets width*height*3 bytes */
int32_t rt_quantize_rgb8(const rt_v3 *image_data, int64_t n_pixels, uint8_t *rgb8);

/*
 * Builds the hierarchy RT_FETCH_BVH traverses (host only, no GPU needed) and checks its
 * invariants for all eight traversal orders: every sphere in exactly one leaf and strictly
 * inside that leaf's box, every box inside its ancestors', skip links forward and leaves of at
 * most 4 spheres in list order.  No counterpart in the reference (its closest-hit loop is the
 * list, main.odin:536-546; the hierarchy is the TODO at :559).  n_nodes / max_leaf may be NULL.
 */
int32_t rt_bvh_validate(const rt_sphere *spheres, int64_t n_spheres, int64_t *n_nodes, int64_t *max_leaf);

/* the counter RNG the kernels use: u32 draw for (seed, pixel, sample, bounce, draw) */
uint32_t rt_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t draw);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
