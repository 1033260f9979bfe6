/* oracle/restate.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement of the reference's per-ray path (oracle/restate.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product library never does.
 *
 * Parity status: PINNED.  The restatement is checked against dumps of the real reference
 * (oracle/_ref/ref_driver, built from /root/reference/src by oracle/Makefile) committed under
 * tests/golden/ -- hit mask, object id, t and float RGB for scenes a-l, statistics for scene m.
 */
#ifndef ORACLE_RESTATE_H
#define ORACLE_RESTATE_H

#include <stddef.h>
#include <stdint.h>
#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* seed == 0: area lights draw from libc rand() exactly like arealight.cpp:5 (single-threaded only);
 * seed != 0: the counter-based generator documented in restate.cpp (identical to the device's). */

/* trace() over n rays (raytracer.cpp:307-329); rgb gets background_colour on a miss (q1.cpp:143-145).
 * counts[0] += BVH::findNearest calls, counts[1] += BVH::calcShadow calls.  threads <= 1: serial. */
int oracle_trace_batch(const rt_flat_scene* scene, const float* e, const float* s, size_t n, int max_depth, uint32_t seed,
                       uint8_t* hit, int32_t* obj_id, float* t, float* rgb, uint64_t counts[2], int threads);

/* BVH::findNearest (bvh.cpp:72-92) */
int oracle_intersect_batch(const rt_flat_scene* scene, const float* e, const float* d, size_t n,
                           uint8_t* hit, int32_t* obj_id, float* t);

/* shadowRay / BVH::calcShadow (raytracer.cpp:111-113, bvh.cpp:123-160) */
int oracle_shadow_batch(const rt_flat_scene* scene, const float* point, const float* light_pos, size_t n,
                        uint8_t* lit, float* shadow);

/* display()'s pixel loop (q1.cpp:130-147) with an aa_x x aa_y sub-pixel grid (1x1 and 2x2 are the reference's);
 * rgb row 0 = bottom row. */
int oracle_render(const rt_flat_scene* scene, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth, uint32_t seed,
                  float* rgb, uint64_t counts[2], int threads);

/* primary ray targets s(x,y) / s_aa (q1.cpp:51-64) for sample k of the n x n grid, into out[3] */
/* Number of exit searches of transmitted rays (Object::transmitRay, objects.cpp:119-125) that found no exit since the last reset: there the
 * reference goes on from Object::cachedHitpoint as an EARLIER ray left it, which no stateless program can follow; the restatement (and the
 * device) use a fixed fallback.  Rays that ran into one are the only ones allowed to differ from the compiled reference. */
uint64_t oracle_state_cases(int reset);

/* TextureTriangle::lightPoint writes the texel into the triangle's own material (texturemesh.cpp:49-50) and Object::lightPoint reads the member
 * AFTER it has traced the reflected ray (objects.cpp:62-79): when that ray's subtree comes back to the same triangle, the outer hit is lit with
 * the texel of the last hit on it.  The restatement reproduces this (it is state of one ray tree only); oracle_texel_cases counts the hits it
 * changed since the last reset, and oracle_texel_state(0) switches it off -- every hit lit with its own texel, which is what the device's
 * level-by-level pipeline computes -- returning the previous setting. */
uint64_t oracle_texel_cases(int reset);
int oracle_texel_state(int on);

void oracle_primary_target(const rt_camera* camera, int width, int height, int x, int y, int aa_x, int aa_y, int k, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
