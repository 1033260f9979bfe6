/* rt_capi.h -- the C-ABI of libraytracer_amd.so: plain pointers and sizes, no C++ or torch types.
 *
 * The reference has no FFI layer: its hot path sits behind three free C++ functions and two globals
 * (/root/reference/src/raytracer.h:12-18).  This header is what a binding of that API onto the MI355X path
 * calls.  Every entry point names the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success and a negative RT_E* code on failure; rt_last_error() gives the
 *     text (thread-local).  Nothing calls exit() (the reference does, raytracer.cpp:127-130).
 *   - "_dev" pointers are DEVICE (HBM) addresses on the scene's GPU; all others are host addresses.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); work is enqueued on it and the
 *     call returns after it has completed unless the function says otherwise.
 *   - vec3 arrays are tightly packed xyz floats (layout-compatible with glm::vec3[], raytracer.h:9-10).
 *   - one rt_scene per (scene, device); calls on DIFFERENT handles may run concurrently, calls on one handle
 *     must come from one thread at a time (the handle owns its queues); rt_render_submit / rt_render_wait keep
 *     several frames of one handle in flight.
 */
#ifndef RT_CAPI_H
#define RT_CAPI_H

#include <stddef.h>
#include <stdint.h>
#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK            0
#define RT_EINVAL       -1   /* bad argument */
#define RT_EIO          -2   /* file missing / unreadable  (reference: exit(EXIT_FAILURE), raytracer.cpp:127-130) */
#define RT_EPARSE       -3   /* malformed scene            (reference: uncaught nlohmann exception)              */
#define RT_EHIP         -4   /* HIP runtime error, no device, kernel image missing */
#define RT_ENOMEM       -5
#define RT_ELIMIT       -6   /* scene exceeds a device-side capacity (e.g. RT_MAX_CSG_LEAVES) */

typedef struct rt_scene rt_scene;   /* device-resident scene + work queues */

/* per-call counters (all optional, all totals over the call) */
typedef struct rt_stats {
	uint64_t rays_nearest;     /* BVH::findNearest calls   (bvh.cpp:72)  : primary + reflection + transmission */
	uint64_t rays_shadow;      /* BVH::calcShadow calls    (bvh.cpp:123) */
	uint64_t nodes_nearest;    /* 64 B BVH nodes read by the nearest-hit kernel   (only when RT_FLAG_COUNTERS) */
	uint64_t prims_nearest;    /* primitive rayhit evaluations in it              (only when RT_FLAG_COUNTERS) */
	uint64_t nodes_shadow;     /* ... and the same two for the shadow kernels */
	uint64_t prims_shadow;
	uint64_t levels;           /* wavefront iterations executed (ray-tree depth reached + 1) */
	uint64_t batch_splits;     /* times a pixel batch was re-run in halves because a ray queue overflowed */
	float    gpu_ms;           /* hipEvent time of the whole call on `stream`, upload/download excluded */
	float    nearest_ms;       /* hipEvent time spent in the nearest-hit traversal kernel (dominant kernel) */
	float    shadow_ms;        /* hipEvent time spent in the shadow traversal kernel */
	float    max_walk;         /* longest nearest-hit walk of the call in nodes (only when RT_FLAG_COUNTERS) */
	uint64_t steps_nearest;    /* node steps of the PACKET walks (one per wave and node visited, x 64 rays): nodes / (64 * steps) is the */
	uint64_t steps_shadow;     /* fraction of a packet's lanes whose own ray is in the node (only when RT_FLAG_COUNTERS)              */
	uint64_t state_cases;      /* exit searches of transmitted rays that found no exit (objects.cpp:123-125): the reference goes on from its */
	                           /* objects' hit caches there -- for top-level meshes state left by EARLIER pixels, which the device does not   */
	                           /* have.  An upper bound on the rays (pixels) that may differ from the reference for that reason; always filled */
} rt_stats;

#define RT_FLAG_COUNTERS   1u   /* count nodes / primitive tests in the traversal kernels (slower) */
#define RT_FLAG_TIME_KERNELS 2u /* bracket each traversal launch with hipEvents to fill nearest_ms / shadow_ms */
#define RT_FLAG_WHOLE_FRAME 4u  /* rt_render_submit only: render this frame on ONE lane instead of cutting it into a shard per lane; frames
                                   submitted back to back are dealt to the lanes in turn.  For callers that keep several frames in flight
                                   and care for frames per second, not for the latency of one frame: a launch over a whole frame amortises
                                   the tail of its longest walks over four times the rays (C2: +11 % with three frames in flight; a frame
                                   alone takes 1.0 ms this way against 0.77 ms in shards).  Same pixels, bit for bit. */

/* ------------------------------------------------------------------------------------------------------
 * Host-side scene handling (no GPU needed).
 * ---------------------------------------------------------------------------------------------------- */

/* Restatement of choose_scene() (raytracer.cpp:117-305) that returns the scene as flat arrays instead of
 * filling globals: parses `json_path`, loads textures / bump maps from `asset_dir` + the file names in the
 * scene (the reference prefixes "scenes/", raytracer.cpp:22,206,232), builds the reference's BVH
 * (bvh.cpp:6-70) and flattens it.  The result owns its arrays; release with rt_flat_scene_free. */
int rt_load_scene_json(const char* json_path, const char* asset_dir, rt_flat_scene** out);
/* Same, from JSON text already in memory. */
int rt_load_scene_json_text(const char* json_text, size_t len, const char* asset_dir, rt_flat_scene** out);
/* Binary form of the flat scene (little-endian dump of the arrays); the additive large-mesh ingest path of
 * SURVEY.md section 8f-2, also how the golden scenes travel to the GPU box. */
int rt_flat_scene_save(const rt_flat_scene* scene, const char* path);
int rt_flat_scene_load(const char* path, rt_flat_scene** out);
void rt_flat_scene_free(rt_flat_scene* scene);

/* Camera basis exactly as setFacing() leaves it (q1.cpp:30-44); rotX/rotY are q1.cpp's rotationX/rotationY. */
int rt_camera_setup(double fov, int width, int height, const float eye[3], float rotX, float rotY, rt_camera* out);

/* RGB float -> 8-bit: clamp to [0,1], c*255+0.5 truncated, NaN -> 0.  `rgb` is W*H*3 with row 0 = BOTTOM
 * row (the GL convention of q1.cpp:125-147); the BMP is written top row first.  The reference never writes
 * a file; the layout is EasyBMP's BMP::WriteToFile (EasyBMP.cpp:424). */
int rt_write_bmp(const char* path, const float* rgb, int width, int height);
int rt_quantize_rgb8(const float* rgb, int width, int height, uint8_t* rgb8_top_down);

/* ------------------------------------------------------------------------------------------------------
 * Device side.
 * ---------------------------------------------------------------------------------------------------- */

/* Number of usable GPUs (0 if none / no HIP runtime). */
int rt_device_count(void);

/* Copies the flat scene to HBM on `device` and allocates the handle's ray/hit queues.  Takes the place of the
 * tail of choose_scene(), which leaves the scene in process globals (raytracer.cpp:29-32,304). */
int rt_scene_upload(const rt_flat_scene* scene, int device, rt_scene** out);
void rt_scene_free(rt_scene* scene);

/* Renders a frame: the display() pixel loop (q1.cpp:130-147) over trace() (raytracer.cpp:307-329), i.e.
 * generate -> [intersect -> shade -> shadow]* -> resolve on the GPU.
 *   aa_x, aa_y    sub-pixel sample grid: 1x1 = s(x,y) (q1.cpp:51-53), 2x2 = the reference's four s_aa samples
 *                 (q1.cpp:55-64); in general aa_x*aa_y samples at ((i+.5)/aa_x, (j+.5)/aa_y), averaged (q1.cpp:140)
 *   max_depth     MAX_REFLECTIONS (raytracer.h:7); the reference value is 16
 *   tile_rank, tile_ranks   render only the 32x8-pixel tiles t with t % tile_ranks == tile_rank (multi-GPU
 *                 sharding); the other pixels of rgb_dev are left untouched.  (0,1) = whole frame.
 *   seed          seeds the counter-based generator that replaces rand() in arealight.cpp:5
 *   rgb_dev       DEVICE buffer, width*height*3 floats, row 0 = bottom row; miss = background colour
 * Enqueues on `stream`, waits for completion, fills *stats if not NULL. */
int rt_render(rt_scene* scene, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth,
              int tile_rank, int tile_ranks, uint32_t seed, uint32_t flags, float* rgb_dev, void* stream, rt_stats* stats);
/* The same frame, asynchronously.  New: the reference renders one pixel at a time on one thread; this is the form a
 * caller that streams frames (the reference's GLUT idle loop, a camera path, one frame per GPU rank) uses to keep
 * the GPU full.  rt_render_submit queues the frame and returns at once with a ticket; rt_render_wait blocks until that
 * frame is complete in rgb_dev, makes `stream` (the one given at submit) wait for it, and fills *stats.  rt_render is
 * submit + wait.  Inside, a frame is cut into interleaved tile shards that run on the handle's own HIP streams
 * ("lanes", RT_LANES, default 4), each driven by its own host thread; shards never wait for each other, so the long
 * tail of one shard's kernels is filled by the next shard -- of the same frame or of the next frame in flight.
 *   - at most 8 tickets may be outstanding; tickets are waited for in any order, each exactly once
 *   - frames in flight must write to DIFFERENT rgb_dev buffers
 *   - a frame with another size / tile split than the ones in flight, or with RT_FLAG_COUNTERS / RT_FLAG_TIME_KERNELS,
 *     first lets the lanes run dry (it is still correct, only not overlapped)
 *   - RT_FLAG_WHOLE_FRAME frames (one lane each, their own accumulators) overlap with any other frame
 *   - the other entry points of this header wait for outstanding frames before they run */
int rt_render_submit(rt_scene* scene, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth,
                     int tile_rank, int tile_ranks, uint32_t seed, uint32_t flags, float* rgb_dev, void* stream, uint64_t* ticket);
int rt_render_wait(rt_scene* scene, uint64_t ticket, rt_stats* stats);
/* Convenience: rt_render into a HOST buffer (allocates a temporary device frame). */
int rt_render_host(rt_scene* scene, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth,
                   uint32_t seed, uint32_t flags, float* rgb_host, rt_stats* stats);

/* trace() for n rays at once (raytracer.h:16, raytracer.cpp:307-329): ray i runs from e[i] towards s[i]
 * (direction s-e, NOT normalised).  Outputs (each may be NULL): hit[i] = trace()'s return value,
 * obj_id[i] = id of hitObject (-1 on a miss), t[i] = its ray parameter, rgb[i] = colour, with the caller's
 * `background_colour` substitution on a miss already applied (q1.cpp:143-145).  Host buffers. */
int rt_trace_batch(rt_scene* scene, const float* e, const float* s, size_t n, int max_depth, uint32_t seed, uint32_t flags,
                   uint8_t* hit, int32_t* obj_id, float* t, float* rgb, rt_stats* stats);

/* trace(e, s, colour, pick = true) (raytracer.cpp:307-327): the one ray is traced like rt_trace_batch traces it, and what the
 * reference would narrate on std::cout on the way -- "object ... hit at", "reflection:", "interior reflection at",
 * "exit object at", ... (objects.cpp:54-140) -- comes back as events, in no particular order; rt_pick_narration puts them into
 * the reference's order (depth first: a hit, its reflected ray, its transmitted ray) and words.  `n_events` receives the
 * number of events the ray tree produced; more than `cap` means the list was cut. */
typedef struct rt_pick_event {
	uint32_t path;      /* the ray's position in the ray tree: 1 = the picked ray, 2p = the reflection of p, 2p + 1 = its transmission */
	int32_t  kind;      /* RT_PICK_* */
	int32_t  arg;       /* RT_PICK_HIT: RT_PRIM_* type of the object; RT_PICK_INTERIOR: number of the internal reflection (0 ..) */
	int32_t  flags;     /* RT_PICK_HIT: bit 0 = the material reflects, bit 1 = it transmits */
	float    p[3];      /* hit point / reflection point / exit point */
} rt_pick_event;
enum { RT_PICK_HIT = 1,        /* "object <type> hit at {p}"                          raytracer.cpp:323-324 */
       RT_PICK_MISS = 2,       /* the ray `path` (> 1) hit nothing: "no additional objects hit"   objects.cpp:71-72, 97-98 */
       RT_PICK_MAXDEPTH = 3,   /* the ray `path` was not traced: "Maximum number of reflections reached."  raytracer.cpp:308-313 */
       RT_PICK_INTERIOR = 4,   /* "interior reflection at {p}"                        objects.cpp:129-131 */
       RT_PICK_EXIT = 5,       /* "exit object at {p}"                                objects.cpp:90-92 */
       RT_PICK_LOST = 6 };     /* "ray lost due to too many internal reflections"     objects.cpp:100-101 */
int rt_trace_pick(rt_scene* scene, const float* e, const float* s, int max_depth, uint32_t seed,
                  uint8_t* hit, float* rgb, rt_pick_event* events, size_t cap, size_t* n_events);
/* The narration as text, one line per std::endl of the reference.  Returns the length of the whole text (without the
 * terminating NUL); at most cap - 1 characters are written to buf. */
size_t rt_pick_narration(const rt_pick_event* events, size_t n, char* buf, size_t cap);

/* BVH::findNearest() for n rays (bvh.h:23, bvh.cpp:72-92): origin e[i], direction d[i] as given. */
int rt_intersect_batch(rt_scene* scene, const float* e, const float* d, size_t n, uint32_t flags,
                       uint8_t* hit, int32_t* obj_id, float* t, rt_stats* stats);

/* NO COUNTERPART IN THE REFERENCE.  rt_intersect_batch for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels and same results as rt_intersect_batch, but
 *   - inputs and outputs are DEVICE buffers of the caller (e_dev, d_dev: n*3 floats; hit_dev n bytes, obj_id_dev n int32,
 *     t_dev n floats; each output may be NULL),
 *   - the ray queue, hit queue and counters the kernels work in are a file-scope device array of the library
 *     (RT_STATIC_MAX_RAYS slots, 3.7 MB, held by every GPU the library runs on until the process ends): the call makes no
 *     hipMalloc / hipFree and no allocating library call,
 *   - n > RT_STATIC_MAX_RAYS is refused with RT_ELIMIT (split the batch),
 *   - the work is enqueued on `stream` and NOT waited for: results are there once `stream` has reached this point.
 * Every call on one GPU -- whatever the rt_scene handle -- shares that GPU's array: calls that overlap in time on one GPU
 * (two streams, two threads) race.  Calls one after the other on one stream are fine; the array keeps the previous call's
 * contents and the call clears what must start at zero.
 * Under stream capture (hipStreamBeginCapture on `stream`): the call records one memset and three kernel launches and does
 * nothing else -- which needs (1) no frame of this handle in flight or un-waited-for (outside capture the call waits for them
 * on the host; a capturing stream cannot), and (2) one earlier, un-captured call on the same handle (the first call loads the
 * kernels and asks for their occupancy).  Either condition missing: RT_EINVAL, nothing recorded.  Replays of the graph follow the
 * same rule as calls: one at a time per GPU. */
#define RT_STATIC_MAX_RAYS 65536u
int rt_intersect_batch_static(rt_scene* scene, const float* e_dev, const float* d_dev, size_t n,
                              uint8_t* hit_dev, int32_t* obj_id_dev, float* t_dev, void* stream);

/* shadowRay() for n queries (raytracer.h:18, raytracer.cpp:111-113 -> BVH::calcShadow, bvh.cpp:123-160):
 * lit[i] = return value, shadow[i] = the attenuation colour on return (starts at (1,1,1)). */
int rt_shadow_batch(rt_scene* scene, const float* point, const float* light_pos, size_t n, uint32_t flags,
                    uint8_t* lit, float* shadow, rt_stats* stats);

/* Multi-GPU frame assembly on rank 0: copies the pixels of the tiles owned by `tile_rank` from a rank's
 * dense frame `src_dev` into `dst_dev` (both width*height*3 floats on the scene's device).  New: the
 * reference is single-process; this is the de-interleave after the RCCL gather. */
int rt_tiles_merge(rt_scene* scene, const float* src_dev, float* dst_dev, int width, int height, int tile_rank, int tile_ranks, void* stream);
/* Packs / unpacks the owned tiles of a frame into a dense buffer of rt_tiles_count()*256*3 floats (gather payload).
 * rt_tiles_merge / _pack / _unpack enqueue one kernel on `stream` and return without waiting for it (stream order is the
 * only synchronisation: the collective that follows is queued behind it, and the host stays free to submit frames). */
int64_t rt_tiles_count(int width, int height, int tile_rank, int tile_ranks);
int rt_tiles_pack(rt_scene* scene, const float* frame_dev, float* packed_dev, int width, int height, int tile_rank, int tile_ranks, void* stream);
int rt_tiles_unpack(rt_scene* scene, const float* packed_dev, float* frame_dev, int width, int height, int tile_rank, int tile_ranks, void* stream);

/* The same exchange with 8-bit pixels: 3 B instead of 12 B per pixel cross xGMI.  Quantised on the device
 * exactly as rt_quantize_rgb8 does on the host (EasyBMP's byte order is applied by the BMP writer, not here).
 *   rt_tiles_pack_rgb8      float frame (row 0 = bottom) -> rt_tiles_count()*256*3 BYTES, the owned tiles in tile order
 *   rt_tiles_unpack_rgb8    such a payload -> the 8-bit frame frame8_dev (width*height*3 bytes, row 0 = TOP as rt_quantize_rgb8 writes)
 *   rt_tiles_quantize_rgb8  float frame -> 8-bit frame for the owned tiles only (the root's own share; (0, 1) = the whole frame)
 * Enqueued on `stream`, not waited for, like rt_tiles_pack. */
int rt_tiles_pack_rgb8(rt_scene* scene, const float* frame_dev, uint8_t* packed_dev, int width, int height, int tile_rank, int tile_ranks, void* stream);
int rt_tiles_unpack_rgb8(rt_scene* scene, const uint8_t* packed_dev, uint8_t* frame8_dev, int width, int height, int tile_rank, int tile_ranks, void* stream);
int rt_tiles_quantize_rgb8(rt_scene* scene, const float* frame_dev, uint8_t* frame8_dev, int width, int height, int tile_rank, int tile_ranks, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * The reference's interactive front end without its window (q1.cpp:23-29, 107-288).
 * A view holds what the GLUT callbacks keep in globals: eye, rotationX / rotationY, the antialias flag, the window size, and the
 * camera basis of the frame last displayed (which is what the movement keys move along, as in the reference).
 *   rt_view_key       keyboard() q1.cpp:179-247: w s a d (move), q e r f (turn, r / f limited to +-pi/2), t g (up / down), l (4x
 *                     supersampling on / off), p (prints camera and facing), space, Escape.  Returns 1 if the key starts a new frame
 *                     (drawing_y = 0), 0 if not, -1 for Escape (the reference exits), -2 for a NULL view (rt_last_error says so); what the
 *                     reference prints goes to `text`.
 *   rt_view_reshape   reshape() q1.cpp:278-288
 *   rt_view_render    display() q1.cpp:107-175 as ONE frame on the GPU: setFacing(), then every pixel (4 s_aa samples with `l` on);
 *                     rgb_dev as in rt_render
 *   rt_view_scanline  the row display() draws in pass number `pass` (every 16th row, 16 sweeps: q1.cpp:124), for a front end that
 *                     reveals the finished frame in the reference's order
 *   rt_view_click     mouse() q1.cpp:251-269: left button down at window position (x, y from the top): the narration of the picked
 *                     ray followed by the HIT / MISS lines, as the reference prints them
 *   rt_view_camera / rt_view_state   the pose the next frame will use / the raw state
 * `scene` may be NULL for a view that is only asked about keys and poses (no GPU needed). */
typedef struct rt_view rt_view;
int rt_view_create(rt_scene* scene, double fov, int width, int height, rt_view** out);
void rt_view_free(rt_view* view);
int rt_view_key(rt_view* view, unsigned char key, char* text, size_t cap, size_t* text_len);
int rt_view_reshape(rt_view* view, int width, int height);
int rt_view_camera(const rt_view* view, rt_camera* camera, int* aa);
int rt_view_state(const rt_view* view, float eye[3], float* rotation_x, float* rotation_y, int* width, int* height, int* needs_redraw);
int rt_view_render(rt_view* view, float* rgb_dev, void* stream, rt_stats* stats);
int rt_view_render_host(rt_view* view, float* rgb_host, rt_stats* stats);   /* convenience, like rt_render_host */
int rt_view_scanline(int height, int pass);
int rt_view_click(rt_view* view, int x, int y_from_top, char* text, size_t cap, size_t* text_len);

const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif
