// rt_kernels.h -- what rt_capi_device.hip (host orchestration) and rt_kernels.hip (device code) share.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_scene.h"
#include "rt_capi.h"
#include "rt_device.h"

#define RT_BLOCK 256          /* threads per workgroup = 4 wavefronts */
#ifndef RT_PACKET_SCALAR_LOADS
#define RT_PACKET_SCALAR_LOADS 0
#endif
#ifndef RT_REFILL_BELOW
#define RT_REFILL_BELOW 48
#endif
#ifndef RT_PACKET_WAVES_PER_SIMD
#define RT_PACKET_WAVES_PER_SIMD 5   /* register budget of the packet kernels: 512 / 5 -> 96 VGPRs (6 and 8 spill and are slower) */
#endif
#ifndef RT_SHADOW_WAVES_PER_SIMD
#define RT_SHADOW_WAVES_PER_SIMD 4   /* k_shadow_point_packet: 128 VGPRs (at 96 its triangle early-out spills inside the walk and costs more than it saves) */
#endif
#ifndef RT_SHADOW2_WAVES_PER_SIMD
#define RT_SHADOW2_WAVES_PER_SIMD 3   /* k_shadow_point_packet2 (two rays per lane): 168 VGPRs; it is launched on 3 workgroups per CU anyway */
#endif
#ifndef RT_AREA_WAVES_PER_SIMD
#define RT_AREA_WAVES_PER_SIMD 4   /* register budget of k_shadow_area: 128 VGPRs (3: 10 % slower on C4, 5: spills, 60 % slower) */
#endif
#define RT_MAX_TREE_DEPTH 20   /* deepest reference tree accepted at upload (the reference's build stops at 16, bvh.h:6) */
#define RT_STACK_DEPTH 32     /* traversal stack entries: per lane in LDS for the per-lane walk (32 x 256 x 4 B = 32 KB per
                                 workgroup), per wave for packets.  A nearer-child-first walk of a binary tree needs at most
                                 its depth: the reference tree plus the sub-trees under fat leaves; checked at upload */

// Treelets (packet traversal): the BVH regrouped into records of 16 slots x 64 B = 1 KB.  A NODE slot has the layout
// of rt_bvh_node (two child boxes, two child references) with count[] replaced by aux[]:
//   child >= 0   inner node, value = treelet * 16 + slot; a child in ANOTHER treelet always points at its slot 0
//   child <  0   leaf, ~child = first entry in leaf_prims (as in rt_bvh_node)
//   aux          for a leaf child whose primitives are 1..63 consecutive triangles (a "run"):
//                  embedded run   count << 26 | RT_TREELET_EMBED | level << 4 | slot : its triangles sit in TRIANGLE
//                                 slots slot .. slot+count-1 of this same treelet (`level` = the treelet's level)
//                  external run   count << 26 | index of the first triangle in tris[] (< 2^25), count 1..63; or, with
//                                 those six bits zero, the count alone (< 2^25): a longer run whose triangles have the
//                                 same indices in tris[] as their entries in leaf_prims[] (first = ~child)
//                RT_TREELET_NO_RUN otherwise (the leaf is then read through leaf_prims)
// A TRIANGLE slot is a copy of the rt_tri record with `mesh` replaced by the triangle's index in tris[] and `uv` by its
// tie-break key (position in leaf_prims + 1).
// Node slots come first; a treelet is filled breadth-first from its root, a node joining only if its own embeddable
// leaves fit as well, so a whole bottom subtree (nodes + triangles) arrives with one fetch.
#define RT_TREELET_SLOTS 16
#define RT_TREELET_MAX_LEVELS 12          /* deeper cuts are rebuilt without embedding, which needs at most 5 */
#define RT_TREELET_NO_RUN 0xffffffffu
#define RT_TREELET_EMBED (1u << 25)
#define RT_TREELET_RUN_MAX 63u
#define RT_TREELET_EMBED_MAX 8u           /* longest run that is embedded; longer all-triangle leaves get a sub-tree */
#define RT_TREELET_SUBLEAF_MAX 4u         /* triangles per leaf of such a sub-tree */
#define RT_TREELET_CONS (1u << 29)        /* flag on a child reference: the node belongs to a sub-tree, its box tests are conservative */
// Quad nodes (the packet walks' form of the tree, DevScene::quads): 128 B = two consecutive slots (the first one even) of such a
// treelet: float box[4][6], int32 child[4], uint32 aux[4] -- child and aux as above (aux of an inner entry: the treelet level of
// the quad it leads to, i.e. which per-level LDS buffer of the wave holds that treelet), RT_QUAD_EMPTY for an entry that does not exist
// (its box is inverted: nothing passes it).  RT_TREELET_CONS on the reference to a quad: its four boxes are conservative.
#define RT_QUAD_EMPTY 0x7fffffff
#ifndef RT_QUAD_SLOTS
#define RT_QUAD_SLOTS 16                   /* 64 B slots per treelet of the quad form (1 KB: one 16 B-per-lane load of a wave).  32 was measured: the kernels alone take the same time -- the C2 tree keeps its 5 treelet levels -- and the larger LDS footprint costs the pipelined frame 12 % (C5: 10 %) */
#endif
#define RT_QUAD_SLOT_BITS (RT_QUAD_SLOTS == 32 ? 5 : 4)
#define RT_QUAD_TREELET_BYTES (RT_QUAD_SLOTS * 64)
/* aux of an embedded run in the quad form: count << 26 | RT_TREELET_EMBED | level << RT_QUAD_SLOT_BITS | slot */
#define RT_PACKET_STACK_DEPTH 48   /* wave stack entries: a step of the quad walk leaves up to three behind, the tree is at most 15 quads deep */
/* The pushes of the packet walks (RT_Q_PUSH in rt_kernels.hip) and the deferred-CSG slots skip an entry that finds the stack full: a
 * wrong image with no error.  That cannot happen while these two bounds hold, and rt_scene_upload / rtk_shadow_area check them with
 * THESE macros, so the walks and their guards cannot drift apart: a quad step leaves at most three entries behind and then descends,
 * so a tree `quadDepth` quads deep needs 3 * quadDepth + 1 entries; a CSG leaf is met once per packet, so deferral needs one slot
 * per CSG object of the scene. */
#define RT_PACKET_STACK_NEED(quadDepth) (3 * (quadDepth) + 1)
#define RT_PACKET_STACK_FITS(quadDepth) (RT_PACKET_STACK_NEED(quadDepth) <= RT_PACKET_STACK_DEPTH)
#define RT_PACKET_DEFER_FITS(nCsg) ((nCsg) <= RT_PACKET_STACK_DEPTH)
#define RT_PACKET_STACK_BYTES (RT_PACKET_STACK_DEPTH * 32)   /* 32 B per entry: the two-ray shadow walk carries two lane masks */
// LDS of a packet kernel's workgroup: the root treelet once (every packet starts there; fetched when the workgroup
// starts), then per wave one 1 KB buffer per deeper treelet level, one more when the scene has external runs to stage,
// and the stack
#define RT_PACKET_WAVE_LDS_BYTES(levels, staging) ((((levels) > 1 ? (levels) - 1 : 0) + ((staging) ? 1 : 0)) * RT_QUAD_TREELET_BYTES + RT_PACKET_STACK_BYTES)
#define RT_COLD_FLOATS 16                                    /* per lane: two directions, two lengths, two attenuation colours (14 used) */
#define RT_PACKET_COLD_BYTES (RT_COLD_FLOATS * 64 * 4)       /* per wave, k_shadow_point_packet2 only, behind the stack */
#define RT_PACKET_LDS_BYTES(levels, staging) (RT_QUAD_TREELET_BYTES + (RT_BLOCK / 64) * RT_PACKET_WAVE_LDS_BYTES(levels, staging))

// SoA queue: component c of slot i lives at f[c * cap + i]
struct RtQueue { float* f; uint32_t cap; };
enum { RQ_OX, RQ_OY, RQ_OZ, RQ_DX, RQ_DY, RQ_DZ, RQ_WR, RQ_WG, RQ_WB, RQ_PIX, RQ_PATH, RQ_COMPS };   // ray: origin, direction (unnormalised), RGB weight, pixel key, tree position
enum { HQ_PRIM, HQ_T, HQ_TAG, HQ_COMPS };                                                             // hit: primitive reference, t, csg boundary tag
enum { SQ_PX, SQ_PY, SQ_PZ, SQ_NX, SQ_NY, SQ_NZ, SQ_VX, SQ_VY, SQ_VZ, SQ_KDR, SQ_KDG, SQ_KDB, SQ_KSR, SQ_KSG, SQ_KSB,
       SQ_SHIN, SQ_WR, SQ_WG, SQ_WB, SQ_PIX, SQ_PATH, SQ_COMPS };
enum { TQ_PRIM, TQ_PX, TQ_PY, TQ_PZ, TQ_DX, TQ_DY, TQ_DZ, TQ_NX, TQ_NY, TQ_NZ, TQ_REFR, TQ_WR, TQ_WG, TQ_WB, TQ_PIX, TQ_PATH, TQ_COMPS };   // transmission request: hit primitive, point, incoming direction, normal, index of refraction, weight of the transmitted ray                                     // surface: what the light loop needs (p, N, V, kd, ks, shininess, weight)

#ifndef RT_PROF
#define RT_PROF 0
#endif
#define RT_TRACE_LEN 2048
#define RT_COUNTER_RESET_BYTES (8 * 4)   /* 32 B: one fill kernel (24 B took the runtime two) */
struct RtCounters {
	uint32_t next_count;      // rays appended to the next level's queue
	uint32_t surf_count;      // surface records appended
	uint32_t head_nearest;    // work cursors of the persistent kernels
	uint32_t head_shadow;
	uint32_t trans_count;     // transmission requests appended by k_shade, consumed by k_transmit
	uint32_t head_shadow_odd; // cursor of the single-light packet launch that follows the paired one when the lights are odd in number
	uint32_t reserved[2];     // ---- everything above is reset at the start of every level (RT_COUNTER_RESET_BYTES) ----
	uint32_t overflow;        // an append found its queue full (the batch is re-run in halves)
	uint32_t pad;
	uint32_t texrec_count;    // RtTexFix records appended by k_shade since k_texel_reset (all levels of a batch)
	uint32_t state_cases;     // exit searches of k_transmit that found no exit since the call began (rt_stats::state_cases); in the lane's base block
	unsigned long long nodes_nearest, prims_nearest;   // RT_FLAG_COUNTERS: 64 B nodes read / primitive tests in k_intersect
	unsigned long long nodes_shadow, prims_shadow;     // ... and in the shadow kernels
	unsigned long long steps_nearest, steps_shadow;    // RT_FLAG_COUNTERS: node steps of the packet walks (one per wave and node, whatever the number of rays in it)
#if RT_PROF
	unsigned long long trace[2][4][RT_TRACE_LEN];      // RT_PROF builds: step traces of four waves of the two packet kernels
	unsigned long long laneHist[2][8];                 // node steps by number of rays in the node: 1-2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64
	uint32_t waveLog[2][8192][8];                      // per wave: start, end (100 MHz wall clock), longest packet, packets; that packet's union steps, max / sum of per-ray nodes, rays
#endif
};

// what k_publish leaves in host memory for one level (Run::levelsAhead polls seq)
struct RtLevelReport { uint32_t next_count, surf_count, overflow, texrec_count, state_cases, seq; };   // texrec_count: RtTexFix records of the batch up to and including this level; state_cases: failed exit searches of the call so far

// Texel state of textured triangles (texturemesh.cpp:49-50, objects.cpp:62-79): TextureTriangle::lightPoint writes the texel into the
// triangle's own material and Object::lightPoint reads it back only after it has traced the reflected ray, so a hit whose reflection
// subtree comes back to the same triangle is lit with the texel of the LAST hit on it.  A level is shaded before the levels below it
// exist, so every hit on a reflective textured triangle leaves a record here, chained per pixel; when a batch's levels are through,
// k_texel_fix finds the records whose subtree holds a later hit on the same triangle and adds the difference of the ambient and
// diffuse terms (shadow rays re-traced: such hits are rare).  Scenes without reflective textured triangles carry none of it.
struct RtTexRec { uint32_t pixKey, path, tri, next; float c[3], wl[3], p[3], N[3]; };   // 64 B
struct RtTexFix { RtTexRec* rec; uint32_t* heads; uint32_t cap; uint32_t* count; };    // rec == nullptr: off.  heads: one list per pixel, kInactive = empty.  count: records of the batch so far (all levels)

// 32.32 fixed-point accumulators: three planes of npix, plus a per-pixel "saw a NaN/inf" byte
struct RtAccum { unsigned long long* rgb; uint8_t* nonfinite; uint32_t npix; };

using rtd::DevScene;

// the first launch of the calling thread that failed since the last call, "launch of <kernel> failed: <hip error>", or null
const char* rtk_take_launch_error();
void rtk_generate(hipStream_t st, const rt_camera& cam, int W, int H, int aaX, int aaY, int tileRank, int tileRanks, uint32_t r0, uint32_t count, RtQueue rq);
void rtk_load_rays(hipStream_t st, const float* a, const float* b, uint32_t n, int bIsTarget, RtQueue rq, uint32_t pixBase);
// nDev (may be null): where the kernels read the number of items themselves, n then being only a bound (a launch queued before the
// host has seen the count)
void rtk_intersect(hipStream_t st, const DevScene& S, bool csg, bool count, bool coherent, int maxBlocks, RtQueue rq, uint32_t n, RtQueue hq, RtCounters* ctr,
                   const uint32_t* nDev = nullptr);
void rtk_export_hits(hipStream_t st, const DevScene& S, RtQueue hq, uint32_t n, uint8_t* hit, int32_t* id, float* t);
// rt_trace_pick: where the shading and transmit kernels leave the events of the narration (null ev: no narration)
struct RtPick { rt_pick_event* ev; uint32_t* count; uint32_t cap; };
// stateCases: where failed exit searches are counted (rt_stats::state_cases; the lane's base counter block, not the level's)
void rtk_transmit(hipStream_t st, const DevScene& S, bool csg, RtQueue tq, uint32_t maxRequests, RtQueue nq, RtCounters* ctr, uint32_t* stateCases, RtAccum A, uint32_t aaSq, int maxDepth,
                  RtPick pick = RtPick{ nullptr, nullptr, 0 });
void rtk_shade(hipStream_t st, const DevScene& S, bool csg, RtQueue rq, RtQueue hq, uint32_t n, RtQueue nq, RtQueue sq, RtQueue tq, RtCounters* ctr, RtAccum A,
               uint32_t aaSq, int maxDepth, int wantSurfaces, uint8_t* outHit, int32_t* outId, float* outT, RtPick pick = RtPick{ nullptr, nullptr, 0 },
               RtTexFix tex = RtTexFix{ nullptr, nullptr, 0, nullptr }, const uint32_t* nDev = nullptr);
// The arrays of a SMALL scene that the per-lane walks and the shading read, as one device blob (sections 16 B aligned) that k_level copies
// into LDS: assembled by rt_scene_upload when blob + per-lane stack fit RT_SMALL_SCENE_LDS_BYTES.  blob == nullptr: the scene is not small.
enum { RT_SMALL_TREELETS, RT_SMALL_LEAF_PRIMS, RT_SMALL_SPHERES, RT_SMALL_BOXES, RT_SMALL_PLANES, RT_SMALL_MATERIALS, RT_SMALL_LIGHTS, RT_SMALL_CSGS, RT_SMALL_CSG_NODES,
       RT_SMALL_LIGHT_LIST, RT_SMALL_TRIS, RT_SMALL_SECTIONS };
#define RT_SMALL_SCENE_LDS_BYTES (60u * 1024u)
struct RtSmallScene {
	const uint4* blob = nullptr;
	uint32_t bytes = 0;          // of the blob (a multiple of 16)
	uint32_t stackRows = 0;      // entries of the per-lane stack: the tree's depth + 2
	uint32_t off[RT_SMALL_SECTIONS] = {};   // byte offsets of the sections; RT_SMALL_TRIS: 0xffffffff when the triangles did not fit (they stay in memory)
};
// a deeper level in one launch: nearest hits, shading, shadow rays towards the `nLights` point-type lights of lightList, appends (k_level);
// surfaces: 0 no shadow-casting lights, 1 count the hits, 2 also write their surface records (area lights: k_shadow_area follows).
// inlineTransmit: exit searches of transmitted rays too (no CSG object and no triangle of the scene may transmit); otherwise requests
// for k_transmit are written.  report (may be null): the level's counts go to the host from this launch (no k_publish behind it).
void rtk_level(hipStream_t st, const DevScene& S, const RtSmallScene& small, bool csg, bool inlineTransmit, RtQueue rq, uint32_t n, const uint32_t* nDev, RtQueue nq, RtQueue sq, RtQueue tq,
               RtCounters* ctr, RtAccum A, uint32_t aaSq, int maxDepth, int surfaces, const int32_t* lightList, uint32_t nLights, uint32_t* stateCases, RtLevelReport* report, uint32_t seq);
// after the last level of a batch: the corrections for `count` records, then the lists and the record counter are emptied again
void rtk_texel_fix(hipStream_t st, const DevScene& S, bool csg, RtTexFix tex, uint32_t count, RtCounters* ctr, RtAccum A, uint32_t aaSq, uint32_t seed, bool apply);
void rtk_shadow_point(hipStream_t st, const DevScene& S, bool csg, bool count, bool coherent, int maxBlocks, RtQueue sq, uint32_t nsurf, const int32_t* lightList, uint32_t nLights,
                      RtAccum A, uint32_t aaSq, RtCounters* ctr, bool countOnDevice = false, int maxBlocksPairs = 0);
void rtk_shadow_area(hipStream_t st, const DevScene& S, bool csg, int nCsg, bool count, RtQueue sq, uint32_t nsurf, int lightIndex, uint32_t samples, uint32_t seed,
                     RtAccum A, uint32_t aaSq, RtCounters* ctr, bool countOnDevice = false);
void rtk_shadow_query(hipStream_t st, const DevScene& S, bool csg, const float* point, const float* lightPos, uint32_t n, uint8_t* lit, float* shadow);
void rtk_resolve(hipStream_t st, RtAccum A, int W, int H, int tileRank, int tileRanks, uint32_t lp0, uint32_t count, uint32_t aaSq, int linear, float* rgb);
void rtk_tiles_copy(hipStream_t st, const float* src, float* dst, int W, int H, int tileRank, int tileRanks, uint32_t count, int mode);
void rtk_tiles_rgb8(hipStream_t st, const void* src, uint8_t* dst, int W, int H, int tileRank, int tileRanks, uint32_t count, int mode);
void rtk_publish(hipStream_t st, const RtCounters* ctr, const uint32_t* texCount, const uint32_t* stateCount, RtLevelReport* hostMapped, uint32_t seq);
