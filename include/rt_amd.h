/* rt_amd.h — C-ABI of the MI355X-native render path (librt_amd.so).
 *
 * Drop-in boundary for the render path of MuellerNico/DD2360-RayTracing.  The reference has no FFI layer: its
 * boundary is the kernel-launch surface of main.cu plus two host functions.  Every entry point
 * below names the reference interface it replaces (file:line under the reference tree).  POD only, plain
 * pointers and sizes, `int` return (0 = ok, otherwise the hipError_t value, or a negative RT_E* code for argument
 * errors); no exceptions cross the boundary; the caller owns every buffer it passes in; opaque handles are
 * created/destroyed by the matching rt_* calls.  All device work is enqueued on the caller-supplied stream
 * (a hipStream_t passed as void*, NULL = default stream) and is asynchronous unless stated otherwise.
 * Thread-compatible, not thread-safe per handle.
 *
 * There is NO CPU fallback: every compute entry point fails (hipError) when no gfx950 device / code object is
 * available.
 */
#ifndef RT_AMD_H
#define RT_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 6

/* argument errors (negative so they never collide with hipError_t) */
#define RT_EINVAL (-1)
#define RT_ENOMEM (-2)
#define RT_EIO (-3)
#define RT_ENOTSUP (-4)
#define RT_ECOMM (-5)   /* the multi-GPU exchange failed (RCCL error, or the custom gather returned non-zero) */

/* real_t selection — precision_types.h:8 (USE_FP16) */
#define RT_PRECISION_FP32 0
#define RT_PRECISION_FP16 1

/* material tags — lambertian / metal / dielectric of material.h:52,62,76; NONE marks a never-initialised
 * ("ghost") slot of the world list (main.cu:160-190 fills only 4+k*k of NUM_SPHERES slots): never hittable. */
#define RT_MAT_NONE (-1)
#define RT_MAT_LAMBERTIAN 0
#define RT_MAT_METAL 1
#define RT_MAT_DIELECTRIC 2

/* curandState (48 bytes) — the per-pixel RNG state buffer of main.cu:383 keeps this layout. */
typedef struct rt_rand_state {
    uint32_t d, v[5];
    int32_t boxmuller_flag, boxmuller_flag_double;
    float boxmuller_extra;
    uint32_t pad_;
    double boxmuller_extra_double;
} rt_rand_state;

/* sphere (sphere.h:7-15) together with the material its mat_ptr points to (material.h:52-116).
 * In FP16 mode every float holds the exact image of the binary16 value. */
typedef struct rt_sphere {
    float center[3];
    float radius;
    int32_t material;   /* RT_MAT_* */
    float albedo[3];    /* lambertian / metal */
    float param;        /* metal: fuzz (already clamped to <=1, material.h:66); dielectric: ref_idx */
} rt_sphere;

/* camera (camera.h:51-56), same field order. */
typedef struct rt_camera {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius;
} rt_camera;

/* OctNode (acceleration_structure.h:35-39), reference layout, for inspection of the built tree. */
typedef struct rt_octnode {
    int32_t level;
    float aabb[6];      /* x_low,y_low,z_low,x_high,y_high,z_high */
    int32_t children[8];
} rt_octnode;

#define RT_OCTREE_MAX_NODES 585 /* NUMBER_NODES, acceleration_structure.h:13 */

/* hit_record (hitable.h:9-15); mat_ptr becomes the index of the sphere that was hit (-1 = miss). */
typedef struct rt_hit_record {
    float t;
    float p[3];
    float normal[3];
    int32_t sphere;
} rt_hit_record;

typedef struct rt_render_ctx rt_render_ctx; /* per-launch state of rt_render: work counters, scheduling workspace, timing events */
typedef struct rt_multi rt_multi;   /* one frame over the GPUs of a node (one process per GPU) */
typedef struct rt_world rt_world;   /* device-resident scene: what d_list / d_world / d_camera reach (main.cu:393-398) */
typedef struct rt_octree rt_octree; /* Octree (acceleration_structure.h:57-62): host reference layout + device traversal copy */

/* Which pixel tiles of the frame this call covers.  The frame is cut into 8x8-pixel tiles (the reference's
 * block shape, main.cu:351-352), numbered row-major from the bottom-left.  Two forms of a part:
 *  - a RANGE (tile_end > tile_begin): the consecutive tiles [tile_begin, tile_end) — a horizontal band of the image.
 *    rt_split_balanced cuts a frame into nparts such bands of equal predicted cost (what rt_multi_render renders by default);
 *    local tile = tile - tile_begin.
 *  - RUNS (tile_begin == tile_end == 0): tiles dealt to the parts in runs of RT_PART_RUN consecutive tiles: tile t lies in run
 *    r = t / RT_PART_RUN, run r belongs to part (r % nparts) and is that part's (r / nparts)-th run;
 *    local tile = (r / nparts) * RT_PART_RUN + t % RT_PART_RUN.  Part 0 never has fewer tiles than another part.
 *    (C5, the slowest of 8 parts on one MI355X: single tiles 106-110 ms, runs of 64: 100-102; whole frame / 8 = 83.)
 * nparts == 1 without a range: the whole frame, buffers in the reference's row-major layout (pixel_index = j*max_x + i).
 * Otherwise buffers are tile-major and compact: element (local_tile*64 + ly*8 + lx).
 * rt_part_pixels() gives the element count of such a buffer. */
#define RT_PART_RUN 64
typedef struct rt_partition {
    int32_t part, nparts;
    int64_t tile_begin, tile_end;
} rt_partition;

/* ---- library ---------------------------------------------------------------------------------------------- */
int rt_abi_version(void);
/* device_count may be NULL. Returns 0 when at least one gfx950 device is usable. */
int rt_device_check(int* device_count);
const char* rt_error_string(int code);

/* ---- host side: scene definition ------------------------------------------------------------------------- */
/* rand_init<<<1,1>>> — main.cu:78-82: curand_init(1984,0,0). Host-side (the world is generated on the host). */
int rt_rand_init(rt_rand_state* rand_state);

/* create_world<<<1,1>>> — main.cu:146-204.  Fills list[num_spheres] (unfilled slots get RT_MAT_NONE) and *cam,
 * advances *rand_state exactly as the reference's single device thread does; *num_created = 4 + k*k filled slots. */
int rt_create_world(rt_sphere* list, int num_spheres, float sphere_radius, rt_camera* cam, int nx, int ny,
                    rt_rand_state* rand_state, int precision, int* num_created);

/* camera::camera — camera.h:22-44 (vfov in degrees). */
int rt_camera_init(rt_camera* cam, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                   float aspect, float aperture, float focus_dist, int precision);

#define RT_TRAVERSAL_REFERENCE 0
#define RT_TRAVERSAL_FAST 1

/* Describes list + camera for the device: replaces the cudaMalloc'ed d_list/d_world/d_camera (main.cu:393-401).
 * Host-only; the device copy is made by rt_world_upload. */
int rt_world_create(const rt_sphere* list, int num_spheres, const rt_camera* cam, int precision, rt_world** out);
/* Creates the device buffers now (otherwise the first render/trace using the handle does it; call this before
 * capturing launches into a hipGraph, since it allocates). */
int rt_world_upload(rt_world* world);
/* How hitable_list::hit (hitable_list.h:16-31, the path taken when no octree is passed) runs on the device.  Both give the
 * reference's hit records bit for bit (fp32): REFERENCE tests every sphere in list order; FAST (default) tests only the
 * spheres the conservative (x,z) grid of the octree path says the ray can touch (the list seen as one unbounded node:
 * lowest index wins among equal t, exactly like the sequential scan) and falls back to the scan for rays it cannot prove.
 * FP16 worlds, lists of fewer than 64 spheres and lists with more than 64 spheres outside the grid's range always use
 * REFERENCE.  The grid is built by the first call that needs it (a render/trace without an octree, or the info call). */
int rt_world_set_list_traversal(rt_world* world, int mode);   /* RT_TRAVERSAL_REFERENCE | RT_TRAVERSAL_FAST */
/* Arithmetic of the fp32 render kernels on this world.  RT_ARITH_IEEE (default): one IEEE binary32 rounding per operation of the
 * reference source, no FMA contraction — the parity contract.  RT_ARITH_CONTRACT: the same kernels compiled with
 * contraction allowed, as nvcc does by default to the reference (Makefile:9 passes no -fmad=false): a*b+c may become one fma with a
 * single rounding (3 instead of 5 operations a dot product).  Pixels then differ from the parity mode in the last bits, and where a
 * last bit flips a decision (a rejection-loop test, a grazing hit) in whole samples: a TOLERANCE mode — tests/test_gpu_contract.py
 * states the measured bounds — reported separately by bench.py --arith contract, never the default.  rt_render / rt_render_progressive
 * only (rt_trace_rays stays IEEE); RT_ENOTSUP for USE_FP16 worlds (binary16 operations are single instructions either way). */
#define RT_ARITH_IEEE 0
#define RT_ARITH_CONTRACT 1
int rt_world_set_arith(rt_world* world, int mode);
int rt_world_list_accel_info(const rt_world* world, int* enabled, int* grid_dim, float* cell_size, int* grid_entries, int* large_spheres);
/* free_world<<<1,1>>> + cudaFree — main.cu:206-219, :464-466. */
int rt_free_world(rt_world* world);

/* buildOctree — acceleration_structure.h:195-217 (host, serial); the upload of main.cu:413-417 is rt_octree_upload.
 * spheres_per_leaf is SPHERES_PER_LEAF (acceleration_structure.h:15, reference value 30). */
int rt_build_octree(const rt_sphere* list, int num_hitables, int spheres_per_leaf, int precision, rt_octree** out);
int rt_octree_upload(rt_octree* octree);   /* the cudaMalloc + cudaMemcpy of main.cu:413-417; implicit on first use */
/* buildOctree + upload in one, ON THE DEVICE: the same tree as rt_build_octree — reference layout (rt_octree_nodes / _leaves),
 * traversal copy and candidate grid, array for array and bit for bit — built from the world's device-resident sphere list
 * (uploads the world if need be) by per-sphere / per-cell kernels and radix sorts; ready to render when the call returns
 * (it synchronises with `stream` a few times: array sizes come back from the device).  N = 100 000, SPHERES_PER_LEAF 320:
 * 1.9 ms instead of 25.6 ms of host build + 3 ms of upload (MI355X box).  USE_FP16 worlds: the tree in binary16 arithmetic, the pair
 * layout and plane table of the binary16 kernels, no candidate grid — again what rt_build_octree + rt_octree_upload give. */
int rt_build_octree_gpu(const rt_world* world, int spheres_per_leaf, rt_octree** out, void* stream);
/* one device-resident array of a tree, copied to the host (parity checks of the two builds; binary16 trees: 0-2 only): 0 traversal nodes, 1 bucket
 * entries (c, r^2), 2 entry -> sphere, 3/4 large spheres + bricks, 5 strip bin starts (columns x fine bins; the x copy, then the z copy), 6/7 strip entries + bricks, 8/9 membership
 * lists, 10 cell -> node, 11/12 membership bitmaps.  *bytes = size of the array; copied when cap suffices. */
int rt_octree_debug_array(const rt_octree* octree, int which, void* out, size_t cap, size_t* bytes);
int rt_free_octree(rt_octree* octree);
int rt_octree_flat_info(const rt_octree* octree, int* n_nodes, int* n_entries);   /* traversal copy: nodes used, bucket entries kept */
/* How hitTree walks the tree on the device.  Both produce the reference's hit records bit for bit (fp32):
 * REFERENCE scans every bucket of every visited level-3 node like traverseTree (acceleration_structure.h:276-304);
 * FAST (default) tests only the spheres a conservative (x,z) grid says the ray can touch — sorted strips: a sphere once
 * per column it overlaps, keyed by the fine bin of its centre; grid_dim = columns per axis, cell_size = column width,
 * grid_entries = registrations of the x copy — and falls back to the scan for rays it cannot prove (DESIGN.md 5.2,
 * App. A).  FP16 trees always use REFERENCE. */
int rt_octree_set_traversal(rt_octree* octree, int mode);
int rt_octree_accel_info(const rt_octree* octree, int* grid_dim, float* cell_size, int* grid_entries, int* large_spheres);
/* reference-layout view of the built tree (for parity checks): counts[0..leafCount), indices[leafCount*spl] */
int rt_octree_info(const rt_octree* octree, int* node_count, int* leaf_count, int* spheres_per_leaf,
                   int* dropped_full, int* dropped_outside);
int rt_octree_nodes(const rt_octree* octree, rt_octnode* out_nodes /* [RT_OCTREE_MAX_NODES] */);
int rt_octree_leaves(const rt_octree* octree, int32_t* counts, int32_t* indices);

/* ---- device side: the hot path ----------------------------------------------------------------------------- */
/* number of elements (pixels) of a buffer for this partition of a max_x x max_y frame */
int64_t rt_part_pixels(int max_x, int max_y, rt_partition part);

/* render_init<<<blocks,threads>>> — main.cu:84-94: curand_init(1984 + pixel_index, 0, 0) per pixel. */
int rt_render_init(int max_x, int max_y, rt_rand_state* d_rand_state, rt_partition part, void* stream);

/* render<<<blocks,threads>>> — main.cu:96-117.  fb: device buffer of vec3 (3 x float, or 3 x binary16 in FP16 mode).
 * d_octree == NULL selects the hitable_list path (USE_OCTREE undefined, main.cu:54). */
int rt_render(void* fb, int max_x, int max_y, int ns, const rt_world* world, rt_rand_state* d_rand_state,
              const rt_octree* d_octree, rt_partition part, void* stream);

/* render_progressive<<<blocks,threads>>> — main.cu:119-142: one sample per call, fb = col (current_sample == 1) or fb += col. */
int rt_render_progressive(void* fb, int max_x, int max_y, int current_sample, const rt_world* world,
                          rt_rand_state* d_rand_state, const rt_octree* d_octree, rt_partition part, void* stream);

/* Render contexts.  rt_render / rt_render_progressive keep their per-launch state (work counters, the scheduling workspace of
 * the pilot pass, timing events) in a context owned by the world handle.  Launches that share a context are ordered by the
 * library (the next call's stream waits for the previous render kernel), so calls on one world from several streams are safe
 * but run one after the other; frames that should overlap on one GPU — two partitions of a frame on two streams — take a
 * context each and go through the *_on entry points.  A context first used inside a hipGraph capture must have been
 * prepared before (rt_render_ctx_reserve, or one uncaptured call of the same frame size — in either precision: the binary16
 * render has a pilot pass and a workspace too); timing events are not recorded during a capture.  A captured rt_render_progressive pass
 * bakes the context's tile-order buffer into the graph: from then on the context refuses (RT_EINVAL) a progressive sequence of a
 * larger frame, which would have to move that buffer — give such a sequence a context of its own for the graph's lifetime. */
int rt_render_ctx_create(rt_render_ctx** out);
int rt_render_ctx_reserve(rt_render_ctx* ctx, int max_x, int max_y, rt_partition part);   /* workspace for frames of this size, now */
int rt_render_ctx_destroy(rt_render_ctx* ctx);
int rt_render_on(rt_render_ctx* ctx, void* fb, int max_x, int max_y, int ns, const rt_world* world, rt_rand_state* d_rand_state,
                 const rt_octree* d_octree, rt_partition part, void* stream);
int rt_render_progressive_on(rt_render_ctx* ctx, void* fb, int max_x, int max_y, int current_sample, const rt_world* world,
                             rt_rand_state* d_rand_state, const rt_octree* d_octree, rt_partition part, void* stream);
int rt_render_ctx_times(rt_render_ctx* ctx, float* ms_out, int max, int* count);           /* as rt_world_render_times */

/* Name of the kernel rt_render (mode 0) / rt_render_progressive (mode 1) launches for this world and tree (d_octree NULL =
 * the hitable_list path), as rocprofv3 shows it without the namespace: "k_render<true,0,4>", "k_render_h<true,0>", ... */
int rt_render_kernel_name(const rt_world* world, const rt_octree* d_octree, int mode, char* out, int cap);

/* Device time of the dominant kernel (k_render / k_render_h) of the most recent rt_render / rt_render_progressive calls on
 * this world: HIP events recorded on the launch stream directly around that kernel (the scheduling pre-pass of rt_render
 * is outside).  Copies up to `max` durations (milliseconds, oldest first, at most the last 64 launches) into ms_out,
 * stores how many in *count, and forgets them.  Synchronises with the recorded events. */
int rt_world_render_times(rt_world* world, float* ms_out, int max, int* count);

/* Scheduling counters of the most recent rt_render on this world / context, read once that launch has finished (waits for it):
 * out4[0] = pixel slots handed out by the work queue, [1] = waves still counted thin (0 after a complete frame), [2] = pixels the
 * pilot pass pre-classified as long chains (0 below 16 samples per pixel; both precisions run the pilot pass), [3] = long-chain
 * handles taken.  Diagnostics only — no counterpart in the reference; which lane renders a pixel never changes the pixel.
 * A render captured into a hipGraph pins ONE slot of the context's counter ring and carries no ordering event: while such a graph
 * replays, its context must not be used by any other launch or replay (they would share the counters, the tile order and the flags) —
 * give every captured render a context of its own (rt_render_ctx_create). */
int rt_world_render_counters(rt_world* world, uint32_t* out4);
int rt_render_ctx_counters(rt_render_ctx* ctx, uint32_t* out4);

/* Reassemble a full row-major frame from nparts tile-major part buffers laid out back to back, each padded to
 * rt_part_pixels(max_x,max_y,{0,nparts}) elements (the layout an all-gather of the parts produces). */
int rt_assemble(void* fb_full, const void* fb_parts, int max_x, int max_y, int nparts, int precision, void* stream);

/* A frame cut into nparts horizontal bands of equal PREDICTED COST (no reference counterpart; extends the launch surface main.cu:422-427).
 * A pilot pass over the whole frame — two one-sample paths per 2x2 pixel block on a private RNG stream, the scheduling pre-pass of
 * rt_render — counts per tile the bounces, the grid entries its paths' walks had to test and the grid columns they stepped through; a tile's
 * cost is a fixed linear form of the three (rt_tuning.h, calibrated on measured band times), and the cuts fall where the running cost passes k/nparts of the total.
 * Integer arithmetic on counts that every GPU of the same kind reproduces bit for bit: every rank of a job computes the same
 * starts[] without talking to the others.  starts[0] = 0 <= ... <= starts[nparts] = number of tiles; part p is
 * rt_partition{p, nparts, starts[p], starts[p+1]} (never empty: RT_EINVAL when the frame has fewer tiles than parts).
 * Why bands: a GPU's rays meet the same part of the scene, as in the undivided frame (runs dealt round-robin keep the whole scene's
 * working set on every GPU for an eighth of the rays).  tile_bounces / tile_tests / tile_columns (host, [tiles], may be NULL) receive the pilot's counts.
 * Synchronises with `stream`.  ctx NULL = the world's own context. */
int rt_split_balanced(rt_render_ctx* ctx, const rt_world* world, const rt_octree* d_octree, int max_x, int max_y, int nparts,
                      int64_t* starts /* [nparts + 1] */, int32_t* tile_bounces, int32_t* tile_tests, int32_t* tile_columns, void* stream);
/* rt_assemble for such a split: band p's buffer begins part_stride_px elements behind band p-1's (>= the largest band). */
int rt_assemble_split(void* fb_full, const void* fb_parts, int max_x, int max_y, int nparts, const int64_t* starts, int64_t part_stride_px,
                      int precision, void* stream);

/* ---- multi-GPU: one frame over the GPUs of one node, one process per GPU -------------------------------------------------
 * No reference counterpart (the reference is single-GPU, launch surface main.cu:422-427).  Every rank computes the same split of
 * the frame's tiles (default RT_SPLIT_BALANCED: rt_split_balanced's bands, recomputed for every frame; RT_SPLIT_RUNS: runs of
 * RT_PART_RUN tiles dealt round-robin, no pilot pass over the whole frame), renders its part into a compact tile-major buffer, and
 * ONE exchange brings the parts to the root (RCCL over xGMI: one ncclGroupStart / ncclRecv x (nranks-1) | ncclSend / ncclGroupEnd,
 * straight from the render buffer into the root's staging slots, on the caller's stream), where rt_assemble / rt_assemble_split
 * writes the row-major frame into fb_full.
 * rt_multi_unique_id: rank 0 creates the 128-byte RCCL id and hands it to the other ranks by any means (a file, MPI,
 * torch.distributed over gloo); rt_multi_init: ncclCommInitRank on the calling process's current device.  RCCL is bound at
 * run time (dlopen): RT_ENOTSUP when it is absent. */
#define RT_MULTI_ID_BYTES 128
int rt_multi_unique_id(void* id_out /* [RT_MULTI_ID_BYTES] */);
int rt_multi_init(rt_multi** out, int rank, int nranks, const void* unique_id);
/* Non-collective check of everything rt_multi_init needs before it enters ncclCommInitRank (RCCL bound with all entry points,
 * a render context and events on the current device): 0, RT_ENOTSUP, or the HIP error.  ncclCommInitRank is collective — a
 * rank that fails early in rt_multi_init leaves its peers blocked inside it — so a job probes on every rank, agrees on the
 * result over its own control plane, and then calls rt_multi_init on all ranks or on none (bench.py does). */
int rt_multi_probe(void);
/* The same with a caller-supplied exchange (MPI, gloo, a test harness) instead of RCCL.  Called on every rank after its part
 * is rendered (enqueued on `stream`): rank r's send_bytes at d_send must arrive at d_parts + r * part_stride_bytes on the
 * root before work enqueued on the root's stream afterwards runs.  The root's own part is in place already (its d_send IS its
 * slot); d_parts is NULL on the other ranks.  Device pointers.  Return 0 on success. */
typedef int (*rt_gather_fn)(void* user, const void* d_send, size_t send_bytes, void* d_parts, size_t part_stride_bytes, int root, void* stream);
int rt_multi_init_custom(rt_multi** out, int rank, int nranks, rt_gather_fn gather, void* user);
int rt_multi_destroy(rt_multi* m);
/* how rt_multi_render divides the frame (the same on every rank): */
#define RT_SPLIT_RUNS 0             /* runs of RT_PART_RUN tiles, round-robin */
#define RT_SPLIT_BALANCED 1         /* bands of equal predicted cost, from a pilot pass over the whole frame on every rank, every frame (default) */
#define RT_SPLIT_BALANCED_CACHED 2  /* ... kept while world, tree, frame size stay the same (a static scene rendered again and again) */
int rt_multi_set_split(rt_multi* m, int mode);
/* the split of the last rt_multi_render: starts[nranks + 1] (RT_SPLIT_RUNS: RT_EINVAL) */
int rt_multi_last_split(rt_multi* m, int64_t* starts);
/* buffers for frames of this size now (otherwise the first rt_multi_render of a larger frame allocates) */
int rt_multi_reserve(rt_multi* m, int max_x, int max_y, int precision, int root);
/* render_init + render of this rank's tiles, the exchange, and on the root the assembled frame in fb_full (device buffer of
 * max_x*max_y vec3, reference layout; ignored on the other ranks).  precision must be the world's (RT_EINVAL otherwise: it sizes
 * the part buffers and the exchange).  Asynchronous on `stream`. */
int rt_multi_render(rt_multi* m, void* fb_full, int max_x, int max_y, int ns, const rt_world* world, const rt_octree* d_octree,
                    int precision, int root, void* stream);
/* device time of this rank's own share of the last rt_multi_render: render_init + render (call_ms) and the render kernel
 * alone (kernel_ms) — the ranks' values side by side show the load balance of the tile split.  Synchronises. */
int rt_multi_last_render_ms(rt_multi* m, float* call_ms, float* kernel_ms);
/* RCCL contexts only: one ncclSend + ncclRecv of `bytes` bytes from this rank to itself inside one group on `stream` */
int rt_multi_selftest(rt_multi* m, const void* d_src, void* d_dst, size_t bytes, void* stream);

/* hitTree (acceleration_structure.h:319-342) / hitable_list::hit (hitable_list.h:16-31) for a batch of rays:
 * d_rays = n x 6 floats (origin, direction) on the device, d_out = n records on the device. t in (0.001, FLT_MAX). */
int rt_trace_rays(const rt_world* world, const rt_octree* d_octree, const float* d_rays, int64_t n,
                  rt_hit_record* d_out, void* stream);

/* ---- host side: output --------------------------------------------------------------------------------------- */
/* output_to_stream — main.cu:321-333: ASCII P3, top row first, int(255.99*c).  fb is a HOST buffer.
 * path == NULL writes to stdout (output mode 0), otherwise to the file (output mode 3 uses "output.ppm"). */
int rt_write_ppm(const char* path, int nx, int ny, const void* fb, int precision);
/* same bytes into memory; returns the length, or the required length when cap is too small / out is NULL */
int64_t rt_format_ppm(int nx, int ny, const void* fb, int precision, char* out, int64_t cap);

/* Binary companions of the ASCII P3 writer (at 4K the P3 text is ~100 MB and dominates end-to-end time).
 * RT_IMAGE_P6: "P6" binary PPM, same quantisation as main.cu:327-329 (int(255.99*c), clamped to 0..255), top row first.
 * RT_IMAGE_PFM: "PF" float image, little-endian (-1.0 scale), bottom row first (the framebuffer's own row order), the
 * gamma-corrected channel values unquantised.  fb is a HOST buffer in the framebuffer layout of rt_render. */
#define RT_IMAGE_P3 0
#define RT_IMAGE_P6 1
#define RT_IMAGE_PFM 2
int rt_write_image(const char* path, int nx, int ny, const void* fb, int precision, int format);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
