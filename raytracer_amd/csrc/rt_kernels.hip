// rt_kernels.hip -- the wavefront pipeline for gfx950 (MI355X): generate -> [intersect -> shade -> transmit -> shadow]* -> resolve.
//
// The reference evaluates one pixel at a time by mutual recursion (trace -> findNearest -> lightPoint -> trace /
// shadowRay).  Here one LEVEL of the ray tree of a whole batch of pixels is processed per
// iteration: rays live in SoA queues in HBM (component c of slot i at base[c*cap + i], so a wave's 64 lanes read
// 256 contiguous bytes per component), child rays are appended to the next level's queue through one atomic per
// wave or workgroup (ballot + prefix popcount), and the recursion's colour arithmetic becomes a per-ray RGB weight: the
// reference's combine  C = (1-kt)*(kr*C_refl + sum lights) + kt*C_trans  (objects.cpp:60-104) is linear in the
// child colours.  Pixel sums are accumulated in 32.32 fixed point with integer atomics, which is associative, so
// the image does not depend on the order in which waves retire.
//
// Kernels (256-thread blocks = 4 waves unless noted):
//   k_generate              q1.cpp:51-64 primary rays, in 8x8-pixel blocks per wave inside 32x8 tiles
//   k_intersect_packet      BVH::findNearest bvh.cpp:72-121 for coherent rays (level 0 of a frame): 64 rays walk the union
//                           of their subtrees over treelets staged in LDS (packetTraverse)
//   k_intersect             ... for incoherent rays (deeper levels, caller-supplied batches): persistent per-lane walk,
//                           per-lane stack in LDS, nearer child first (travRound)
//   k_shade                 Object::lightPoint objects.cpp:54-105 minus the shadow rays: normals (bump.cpp), textures
//                           (texturemesh.cpp), ambient term, reflected children, requests for transmitted children
//   k_transmit              Object::transmitRay objects.cpp:107-140 over the compacted requests (64-thread blocks)
//   k_shadow_point_packet2, Directional/Point/Spot::lightPoint objects.cpp:399-458 -> BVH::calcShadow bvh.cpp:123-160:
//   k_shadow_point_packet,  packets with two lights per surface (two rays per lane), with one, and the per-lane form
//   k_shadow_point
//   k_shadow_area           AreaLight::lightPoint arealight.cpp:7-22, one lane per sample, segmented wave reduction
//   k_texel_fix             the material state TextureTriangle::lightPoint leaves behind (texturemesh.cpp:49-50): hits on reflective
//                           textured triangles that their own reflection subtree comes back to, corrected after the last level
//   k_resolve               fixed point -> float RGB (+ the /4 of q1.cpp:140 for supersampling)
#include "rt_kernels.h"
#include "rt_device.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

using namespace rtd;

#ifndef RT_SCHED_SPLIT
#define RT_SCHED_SPLIT 0
#endif
#ifndef RT_ABLATE
#define RT_ABLATE 0   /* diagnostic builds: bit 0 = walks without their triangle tests (tools/ab_libs.sh) */
#endif

namespace {

constexpr int kBlock = RT_BLOCK;
constexpr int kStack = RT_STACK_DEPTH;
constexpr uint32_t kInactive = 0xffffffffu;

__device__ __forceinline__ float qf(const RtQueue& q, int comp, uint32_t i) { return q.f[(size_t)comp * q.cap + i]; }
__device__ __forceinline__ uint32_t qu(const RtQueue& q, int comp, uint32_t i) { return __float_as_uint(q.f[(size_t)comp * q.cap + i]); }
__device__ __forceinline__ void qsf(const RtQueue& q, int comp, uint32_t i, float v) { q.f[(size_t)comp * q.cap + i] = v; }
__device__ __forceinline__ void qsu(const RtQueue& q, int comp, uint32_t i, uint32_t v) { q.f[(size_t)comp * q.cap + i] = __uint_as_float(v); }

// One slot per lane with the predicate set, one atomic per wave.  Must be reached by every lane of the wave.
__device__ __forceinline__ uint32_t wavePush(bool pred, uint32_t* counter) {
	const unsigned long long mask = __ballot(pred);
	if (mask == 0) return kInactive;
	const uint32_t lane = __lane_id();
	const int leader = __ffsll((long long)mask) - 1;
	uint32_t base = 0;
	if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
	base = __shfl(base, leader);
	return pred ? base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) : kInactive;
}

// 32.32 fixed-point accumulation (associative => order-independent image)
__device__ __forceinline__ void accumChannel(unsigned long long* plane, uint32_t pixel, float c) {
	if (c == 0.0f) return;
	double v = (double)c * 4294967296.0;
	if (v > 4.0e18) v = 4.0e18;
	if (v < -4.0e18) v = -4.0e18;
	atomicAdd(plane + pixel, (unsigned long long)__double2ll_rn(v));
}
__device__ __forceinline__ void accumAdd(const RtAccum& A, uint32_t pixel, f3 c) {
	if (!(isfinite(c.x) && isfinite(c.y) && isfinite(c.z))) { A.nonfinite[pixel] = 1; return; }
	accumChannel(A.rgb, pixel, c.x);
	accumChannel(A.rgb + A.npix, pixel, c.y);
	accumChannel(A.rgb + 2 * (size_t)A.npix, pixel, c.z);
}

// two contributions to one pixel: converted to fixed point one by one, as two accumAdd calls would, added as integers
__device__ __forceinline__ void accumAdd2(const RtAccum& A, uint32_t pixel, f3 c1, bool has1, f3 c2, bool has2) {
	const bool ok1 = has1 && isfinite(c1.x) && isfinite(c1.y) && isfinite(c1.z), ok2 = has2 && isfinite(c2.x) && isfinite(c2.y) && isfinite(c2.z);
	if ((has1 && !ok1) || (has2 && !ok2)) A.nonfinite[pixel] = 1;
	auto fixed = [](float c) -> long long {
		if (c == 0.0f) return 0;
		double v = (double)c * 4294967296.0;
		if (v > 4.0e18) v = 4.0e18;
		if (v < -4.0e18) v = -4.0e18;
		return __double2ll_rn(v);
	};
	const float v1[3] = { c1.x, c1.y, c1.z }, v2[3] = { c2.x, c2.y, c2.z };
	for (int k = 0; k < 3; k++) {
		const long long sum = (ok1 ? fixed(v1[k]) : 0) + (ok2 ? fixed(v2[k]) : 0);
		if (sum != 0) atomicAdd(A.rgb + (size_t)k * A.npix + pixel, (unsigned long long)sum);
	}
}

// One or two contributions per lane, for frames with several samples per pixel: their samples are neighbours in the queues, so sixteen
// lanes of a wave would send their atomics to the same three addresses, which take them one after the other (C5, 16 samples per pixel:
// the shading kernel spent two thirds of its time there).  Lanes of a row of 16 that follow each other with the same pixel add their
// fixed-point values first (integers: the accumulator ends up with the same bits as after the separate atomics) and the last lane of
// each run sends the sum.  Must be reached by every lane of the wave.
__device__ __forceinline__ long long accumFixed(float c) {
	if (c == 0.0f) return 0;
	double v = (double)c * 4294967296.0;
	if (v > 4.0e18) v = 4.0e18;
	if (v < -4.0e18) v = -4.0e18;
	return __double2ll_rn(v);
}
#define RT_ROW_SHR(x, n) __builtin_amdgcn_update_dpp(0, (int)(x), 0x110 + (n), 0xf, 0xf, false)
__device__ __forceinline__ void accumAddRows(const RtAccum& A, uint32_t pixel, f3 c1, bool has1, f3 c2, bool has2) {
	const bool ok1 = has1 && isfinite(c1.x) && isfinite(c1.y) && isfinite(c1.z), ok2 = has2 && isfinite(c2.x) && isfinite(c2.y) && isfinite(c2.z);
	if ((has1 && !ok1) || (has2 && !ok2)) A.nonfinite[pixel] = 1;
	long long v[3];
	v[0] = (ok1 ? accumFixed(c1.x) : 0) + (ok2 ? accumFixed(c2.x) : 0);
	v[1] = (ok1 ? accumFixed(c1.y) : 0) + (ok2 ? accumFixed(c2.y) : 0);
	v[2] = (ok1 ? accumFixed(c1.z) : 0) + (ok2 ? accumFixed(c2.z) : 0);
	const uint32_t lane = __lane_id();
	const uint32_t prev = (uint32_t)RT_ROW_SHR(pixel, 1);
	const unsigned long long heads = __ballot((lane & 15u) == 0u || prev != pixel);   // (rows start runs)
	const uint32_t start = 63u - (uint32_t)__clzll((long long)(heads & ((2ull << lane) - 1ull)));
#pragma unroll
	for (int k = 0; k < 3; k++) {
		uint32_t lo = (uint32_t)v[k], hi = (uint32_t)((unsigned long long)v[k] >> 32);
		// inclusive sum over the lanes of the run up to this one: steps of 1, 2, 4, 8 inside the row
#define RT_ROW_STEP(n) { const uint32_t l2 = (uint32_t)RT_ROW_SHR(lo, n), h2 = (uint32_t)RT_ROW_SHR(hi, n); \
			if (lane >= start + (n)) { const unsigned long long t = (((unsigned long long)hi << 32) | lo) + (((unsigned long long)h2 << 32) | l2); lo = (uint32_t)t; hi = (uint32_t)(t >> 32); } }
		RT_ROW_STEP(1) RT_ROW_STEP(2) RT_ROW_STEP(4) RT_ROW_STEP(8)
#undef RT_ROW_STEP
		v[k] = (long long)(((unsigned long long)hi << 32) | lo);
	}
	const bool tail = lane == 63u || ((heads >> (lane + 1u)) & 1ull) != 0ull;
	if (tail) {
#pragma unroll
		for (int k = 0; k < 3; k++)
			if (v[k] != 0) atomicAdd(A.rgb + (size_t)k * A.npix + pixel, (unsigned long long)v[k]);
	}
}
#undef RT_ROW_SHR

// local pixel number (tile-major, 8x8 blocks inside a 32x8 tile) -> image coordinates
__device__ __forceinline__ bool tilePixel(uint32_t lp, int W, int H, int rank, int ranks, int& x, int& y) {
	const uint32_t tilesX = (uint32_t)(W + 31) / 32, tilesY = (uint32_t)(H + 7) / 8;
	const unsigned long long gt = (unsigned long long)(lp >> 8) * (uint32_t)ranks + (uint32_t)rank;
	if (gt >= (unsigned long long)tilesX * tilesY) return false;
	const uint32_t tx = (uint32_t)(gt % tilesX), ty = (uint32_t)(gt / tilesX);
	const uint32_t within = lp & 255u, sub = within >> 6, l = within & 63u;
	x = (int)(tx * 32 + sub * 8 + (l & 7u));
	y = (int)(ty * 8 + (l >> 3));
	return x < W && y < H;
}

// ---------------------------------------------------------------------------------------------------------
// primitive dispatch for BVH leaves: object->rayhit(e, d)  (bvh.cpp:112,148)
// ---------------------------------------------------------------------------------------------------------
template <bool CSG>
__device__ __forceinline__ float primHit(const DevScene& S, uint32_t ref, f3 e, f3 d, uint32_t& tag) {
	const uint32_t ix = RT_PRIM_INDEX(ref);
	const uint32_t ty = RT_PRIM_TYPE(ref);
	if (ty == RT_PRIM_TRI) return triHit(S.tris + ix, e, d, false);
	if (ty == RT_PRIM_SPHERE) {
		const float4 s = *reinterpret_cast<const float4*>(S.spheres + ix);
		return sphereHit(mk3(s.x, s.y, s.z), s.w, e, d, false);
	}
	if (ty == RT_PRIM_BOX) {
		const float4 a = *reinterpret_cast<const float4*>(S.boxes + ix);
		const float2 b = *reinterpret_cast<const float2*>(S.boxes[ix].b + 4);
		const float bb[6] = { a.x, a.y, a.z, a.w, b.x, b.y };
		return boxHit(bb, e, d, false);
	}
	if (CSG) return csgEval(S, S.csgs[ix], e, d, false, tag);
	return 0.0f;
}
__device__ __forceinline__ int primMaterial(const DevScene& S, uint32_t ref) {
	const uint32_t ix = RT_PRIM_INDEX(ref);
	switch (RT_PRIM_TYPE(ref)) {
	case RT_PRIM_SPHERE: return S.spheres[ix].mat;
	case RT_PRIM_TRI: return S.tris[ix].mat;
	case RT_PRIM_BOX: return S.boxes[ix].mat;
	case RT_PRIM_CSG: return S.csgs[ix].mat;
	case RT_PRIM_PLANE: return S.planes[ix].mat;
	}
	return 0;
}
__device__ __forceinline__ int primId(const DevScene& S, uint32_t ref) {
	const uint32_t ix = RT_PRIM_INDEX(ref);
	switch (RT_PRIM_TYPE(ref)) {
	case RT_PRIM_SPHERE: return S.spheres[ix].id;
	case RT_PRIM_TRI: return S.tris[ix].id;
	case RT_PRIM_BOX: return S.boxes[ix].id;
	case RT_PRIM_CSG: return S.csgs[ix].id;
	case RT_PRIM_PLANE: return S.planes[ix].id;
	}
	return -1;
}

struct NodeRegs { float b0[6], b1[6]; int32_t c0, c1; };
__device__ __forceinline__ NodeRegs loadNode(const rt_bvh_node* n) {   // one 64 B record = 4 x 16 B loads
	const float4* q = reinterpret_cast<const float4*>(n);
	const float4 a = q[0], b = q[1], c = q[2];
	const float2 d = *reinterpret_cast<const float2*>(q + 3);
	NodeRegs r;
	r.b0[0] = a.x; r.b0[1] = a.y; r.b0[2] = a.z; r.b0[3] = a.w; r.b0[4] = b.x; r.b0[5] = b.y;
	r.b1[0] = b.z; r.b1[1] = b.w; r.b1[2] = c.x; r.b1[3] = c.y; r.b1[4] = c.z; r.b1[5] = c.w;
	r.c0 = __float_as_int(d.x); r.c1 = __float_as_int(d.y);
	return r;
}

// ---------------------------------------------------------------------------------------------------------
// Node test.  BoundingBox::intersect (objects.cpp:11-44) divides six times per box; on the device a correctly
// rounded fp32 divide is ~11 VALU instructions, which made the two-box node test ~2/3 of all instructions.
// Here the slabs are first evaluated with one multiply by the ray's precomputed 1/d.  Those t values are within 3 ulp
// of the reference's (the subtraction, a reciprocal good to 1 ulp, the product), so every decision taken with a relative margin of 1e-6
// (8 ulp) is the reference's decision; only when
// tnear / tfar / t_limit are closer than the margin is the exact divide-based test evaluated (rare, divergent:
// 5 380 of 72 M tests on the C2 frame).  The hit condition of the reference, "tnear <= tfar, tfar >= 0, entry <=
// limit" (entry = tnear, or 0 from inside the box), is tnear <= min(tfar, limit) and tfar >= 0; the sign of tfar
// is that of the reference's quotient unless it underflows.  The (min, max) pairs of a box are neighbours in memory,
// so the six subtractions and six multiplications are three packed instructions each.
// A ray with a zero (or denormal) direction component has an infinite 1/d: 0 * inf then yields NaN where the
// reference's d == 0 branch decides, so those rays (`special`) send every box with a NaN slab to the exact test.
// Returns 0 = skip, 1 = visit, 2 = undecided.
// ---------------------------------------------------------------------------------------------------------
typedef float v2f_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int boxClassify(const float* b, const f3& e, const f3& inv, bool special, float tlimit, float& tEntry) {
	const v2f_ bx = { b[0], b[1] }, by = { b[2], b[3] }, bz = { b[4], b[5] };
	const v2f_ tx = (bx - e.x) * inv.x, ty = (by - e.y) * inv.y, tz = (bz - e.z) * inv.z;
	const float lo = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), -RT_MAX_T));
	const float hi = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), RT_MAX_T));
	tEntry = lo;
	if (special) {
		if (tx.x != tx.x || tx.y != tx.y || ty.x != ty.x || ty.y != ty.y || tz.x != tz.x || tz.y != tz.y) return 2;
		if (lo > 3.0e38f || hi < -3.0e38f) return 0;   // origin outside a slab the ray runs parallel to: exact
	}
	const float u = fminf(hi, tlimit);
	const float g = u - lo;
	const float m = 1e-6f * (fabsf(lo) + fabsf(u)) + 1e-30f;
	if (g > m && hi > 1e-30f) return 1;     // certainly hit within the limit
	if (g < -m || hi < -1e-30f) return 0;   // certainly missed, or certainly beyond the limit
	return 2;                               // too close to call (or not a number): the reference's own arithmetic decides
}

// The same test without control flow, for the packet walk (every instruction of a step is paid by the whole wave, and
// branches around a few instructions cost more than they save): `hit` = certainly visit, `amb` = the exact test
// decides.  Lanes of `special` rays are made ambiguous by the caller.
__device__ __forceinline__ void boxClassifyFlat(const float* b, const f3& e, const f3& inv, float tlimit, float& tEntry, bool& hit, bool& amb) {
	const v2f_ bx = { b[0], b[1] }, by = { b[2], b[3] }, bz = { b[4], b[5] };
	const v2f_ tx = (bx - e.x) * inv.x, ty = (by - e.y) * inv.y, tz = (bz - e.z) * inv.z;
	const float lo = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), -RT_MAX_T));
	const float hi = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), RT_MAX_T));
	tEntry = lo;
	const float u = fminf(hi, tlimit);
	const float g = u - lo;
	const float m = 1e-6f * (fabsf(lo) + fabsf(u)) + 1e-30f;
	hit = g > m && hi > 1e-30f;
	amb = !(hit || g < -m || hi < -1e-30f);
}

// ---------------------------------------------------------------------------------------------------------
// Per-lane traversal state for both queries:
//   nearest  BVH::findNearest (bvh.cpp:72-92) + findRecursive (bvh.cpp:94-121)
//   any-hit  BVH::calcShadow (bvh.cpp:123-126) + shadowRecursive (bvh.cpp:128-160)
//
// The reference walks left-then-right and keeps the FIRST of equal-t hits (strict <).  The nearest walk visits the
// nearer child first, which prunes far more, and reproduces the reference's winner by breaking t ties with the
// primitive's position in the reference's DFS leaf order (key = index into leaf_prims + 1; planes, which the
// reference tests before the tree, have key 0).  For the any-hit query the order is free (an opaque occluder ends
// it whatever came before; the attenuation product commutes).  Planes are not in the tree and never occlude.
//
// The walk is organised "while-while": every lane first descends through inner nodes until it holds a leaf (or is
// done), then all lanes test their leaf's primitives together, so box code and primitive code do not alternate
// inside a divergent wave.  `cur` >= 0: inner node, < 0: leaf at leaf_prims[~cur], kEmpty: finished.
// ---------------------------------------------------------------------------------------------------------
constexpr int32_t kEmpty = 0x7fffffff;

struct Trav {
	f3 e, d, inv;
	bool special;   // 1/d is not finite on some axis (boxClassify)
	int32_t cur;
	uint32_t aux, base;   // per-lane walk, when cur is a leaf: its aux word and the treelet it hangs in
	int sp;
	float tmin; uint32_t best, bestKey, bestTag;   // nearest
	f3 shadow; float len; bool lit;                 // any-hit
	unsigned long long cntNodes, cntPrims;
	uint32_t cntSteps;   // packet walks, COUNT builds: node steps of the wave (the same in every lane)
#if RT_PROF
	uint32_t steps;
#endif
};

template <bool ANYHIT, bool RCP = true>
__device__ __forceinline__ void travInit(const DevScene& S, Trav& T, f3 e, f3 d) {
	T.e = e; T.d = d;
	// (1/d only feeds the multiply tests, whose margin of 1e-6 covers a reciprocal good to 1 ulp: v_rcp_f32, one instruction where the
	// correctly rounded divide takes eleven; a zero or denormal component gives an infinite or huge value and `special` sends the ray's
	// boxes to the exact test)
	// (RCP false, the two-ray shadow kernel: with this and the root test below the allocator takes 164 registers instead of 156, and two
	// nearest-hit waves no longer fit on a SIMD beside two of its waves: C2 frame rate -1.6 % with all lanes running)
	T.inv = RCP ? mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)) : mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	T.special = !(fabsf(T.inv.x) <= 3.0e38f && fabsf(T.inv.y) <= 3.0e38f && fabsf(T.inv.z) <= 3.0e38f);
	T.sp = 0; T.cur = kEmpty;
	T.cntNodes = 0; T.cntPrims = 0; T.cntSteps = 0;
#if RT_PROF
	T.steps = 0;
#endif
	T.tmin = RT_MAX_T; T.best = RT_NO_HIT; T.bestKey = 0; T.bestTag = 0;   // bvh.cpp:73
	T.shadow = mk3(1.0f, 1.0f, 1.0f); T.lit = true; T.len = 0;
	if (!ANYHIT) {
		for (int i = 0; i < S.n_planes; i++) {   // bvh.cpp:77-86
			const float4 a = *reinterpret_cast<const float4*>(S.planes + i);
			const float2 b = *reinterpret_cast<const float2*>(S.planes[i].normal + 1);
			const float t = planeHit(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), e, d, false);
			if (t > 0 && t < T.tmin) { T.best = RT_PRIM_REF(RT_PRIM_PLANE, i); T.tmin = t; }
		}
	} else T.len = length(d);
	if (S.root_child < 0 && S.root_count == 0) return;
	// the root's own box (bvh.cpp:97-100 / 131-134): the multiply test the children get, the six exact divides only when it cannot tell
	float tEntry;
	const int cls = boxClassify(S.root_box, e, T.inv, T.special, ANYHIT ? 1.0f : T.tmin, tEntry);
	if (cls == 0) return;
	if (cls == 2) {
		const float tr = boxIntersect(S.root_box, e, d, false);
		if (tr < 0 || tr > (ANYHIT ? 1.0f : T.tmin)) return;
	}
	T.cur = S.treelet_root;   // slot 0 of treelet 0, or the one leaf of a one-leaf tree
	T.aux = S.treelet_root_aux; T.base = 0;
}

// bvh.cpp:109-118 / 145-157 for one candidate hit at parameter t
template <bool ANYHIT>
__device__ __forceinline__ bool travAccept(const DevScene& S, Trav& T, float t, uint32_t ref, uint32_t tag, uint32_t key, int mat) {
	if (!ANYHIT) {
		if ((double)t > 1e-5 && (t < T.tmin || (t == T.tmin && key < T.bestKey))) { T.best = ref; T.bestTag = tag; T.tmin = t; T.bestKey = key; }
	} else if (t < 1.0f && (double)(t * T.len) > 1e-5) {
		const float* kt = S.materials[mat].transmissive;
		if (!isZero3(kt)) T.shadow = T.shadow * ld3(kt);
		else { T.lit = false; return false; }
	}
	return true;
}

// The per-lane walk runs over the same tree as the packets (the treelet records, rt_kernels.h, read from memory one
// 64 B node at a time), so it too finds the triangles of small leaves next to their nodes and walks the conservative
// sub-trees of fat leaves.
template <bool ANYHIT, bool CSG, bool COUNT>
__device__ __forceinline__ void travLeaf(const DevScene& S, Trav& T) {
	const uint32_t info = T.aux;
	if (info == RT_TREELET_NO_RUN) {   // any primitives: through leaf_prims
		int32_t at = ~T.cur;
		bool more = true;
		while (more) {
			const uint32_t entry = S.leaf_prims[at];
			const uint32_t ref = entry & ~RT_PRIM_LAST;
			more = (entry & RT_PRIM_LAST) == 0;
			uint32_t tag = 0;
			const float t = primHit<CSG>(S, ref, T.e, T.d, tag);
			if (COUNT) T.cntPrims++;
			if (!travAccept<ANYHIT>(S, T, t, ref, tag, (uint32_t)at + 1u, ANYHIT ? primMaterial(S, ref) : 0)) more = false;
			at++;
		}
		return;
	}
	const bool longRun = (info >> 26) == 0u;
	const uint32_t count = longRun ? info : info >> 26;
	const bool embedded = !longRun && (info & RT_TREELET_EMBED);
	const uint32_t first = embedded ? 0u : longRun ? (uint32_t)~T.cur : info & (RT_TREELET_EMBED - 1u);
	const rt_tri* tris = embedded ? reinterpret_cast<const rt_tri*>(S.treelets) + (size_t)T.base * RT_TREELET_SLOTS + (info & 15u) : S.tris + first;
	for (uint32_t k = 0; k < count; k++) {
		const float4* q = reinterpret_cast<const float4*>(tris + k);
		const float4 a = q[0], b = q[1], c = q[2], d = q[3];
		float t = 0.0f;   // (what triHit returns when the plane test fails)
		if (!ANYHIT || __ballot(triCandidate<ANYHIT>(mk3(a.x, a.y, a.z), mk3(c.y, c.z, c.w), T.e, T.d)) != 0)
			t = triHit(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), mk3(c.y, c.z, c.w), T.e, T.d, false);
		if (COUNT) T.cntPrims++;
		const uint32_t index = embedded ? __float_as_uint(d.z) : first + k;
		const uint32_t key = embedded ? __float_as_uint(d.w) : (uint32_t)~T.cur + 1u + k;
		if (!travAccept<ANYHIT>(S, T, t, RT_PRIM_REF(RT_PRIM_TRI, index), 0u, key, __float_as_int(d.y))) break;
	}
}

// Stack entries: an inner node's reference (>= 0), or -(2 * parent + k) - 1 for leaf child k of node `parent` (its
// reference and aux word are read from the parent again when it is popped).
__device__ __forceinline__ void travPop(const DevScene& S, int32_t (*stack)[kBlock], Trav& T) {
	if (T.sp == 0) { T.cur = kEmpty; return; }
	const int32_t v = stack[--T.sp][threadIdx.x];
	if (v >= 0) { T.cur = v; return; }
	const uint32_t code = (uint32_t)(-(v + 1));
	const rt_bvh_node* n = reinterpret_cast<const rt_bvh_node*>(S.treelets) + (code >> 1);
	T.cur = n->child[code & 1u]; T.aux = (uint32_t)n->count[code & 1u]; T.base = code >> 5;
}

// One round: descend to the next leaf, test it, pick what comes next.  Returns with T.cur == kEmpty when the query is over.
template <bool ANYHIT, bool CSG, bool COUNT>
__device__ __forceinline__ void travRound(const DevScene& S, int32_t (*stack)[kBlock], Trav& T) {
	const int tid = threadIdx.x;
	while (T.cur >= 0 && T.cur != kEmpty) {
		const bool cons = ((uint32_t)T.cur & RT_TREELET_CONS) != 0u;
		const uint32_t nidx = (uint32_t)T.cur & (RT_TREELET_CONS - 1u);
		const rt_bvh_node* np = reinterpret_cast<const rt_bvh_node*>(S.treelets) + nidx;
		const NodeRegs N = loadNode(np);
		const uint32_t x0 = (uint32_t)np->count[0], x1 = (uint32_t)np->count[1];
		if (COUNT) T.cntNodes++;
		const float tlimit = ANYHIT ? 1.0f : T.tmin;
		float t0, t1;
		int k0 = boxClassify(N.b0, T.e, T.inv, T.special, tlimit, t0);
		int k1 = boxClassify(N.b1, T.e, T.inv, T.special, tlimit, t1);
		if (k0 == 2 || k1 == 2) {   // too close to call with the multiply: a visit in a conservative sub-tree, else the reference's own arithmetic decides
			// Nearest-hit walk: a box whose entry is within the margin of the best hit so far is visited whatever the exact comparison says.
			// The reference prunes a box against the hits it found EARLIER in its own left-to-right order; this walk goes near child first,
			// so its best hit may be one the reference finds later -- typically the other triangle of an edge the ray hits exactly, with
			// the same t bit for bit -- and a box entry an ulp beyond that t (slab and triangle arithmetic round differently) would hide
			// the triangle the reference found first.  Visiting costs a leaf test; the tie is then settled by the keys.
			if (k0 == 2) { if (cons) k0 = 1; else { const float t = boxIntersect(N.b0, T.e, T.d, false); k0 = !(t < 0 || (ANYHIT && t > tlimit)); } }
			if (k1 == 2) { if (cons) k1 = 1; else { const float t = boxIntersect(N.b1, T.e, T.d, false); k1 = !(t < 0 || (ANYHIT && t > tlimit)); } }
		}
		int take = -1;   // the child to continue with
		if (k0 && k1) {
			const bool swap = !ANYHIT && t1 < t0;
			const int far = swap ? 0 : 1;
			const int32_t cf = far ? N.c1 : N.c0;
			if (T.sp < kStack) stack[T.sp++][tid] = cf >= 0 ? cf : -(int32_t)(2u * nidx + (uint32_t)far) - 1;
			take = 1 - far;
		} else if (k0) take = 0;
		else if (k1) take = 1;
		if (take < 0) travPop(S, stack, T);
		else { T.cur = take ? N.c1 : N.c0; T.aux = take ? x1 : x0; T.base = nidx >> 4; }
	}
	if (T.cur != kEmpty) {   // a leaf
		travLeaf<ANYHIT, CSG, COUNT>(S, T);
		if (ANYHIT && !T.lit) { T.cur = kEmpty; T.sp = 0; }
		else travPop(S, stack, T);
	}
}

// ---------------------------------------------------------------------------------------------------------
// Packet traversal for COHERENT rays (camera rays of an 8x8 pixel block, their shadow rays to one light, the samples
// of one area light from one surface point).
//
// Measured on MI355X (tools/ubench/gather*.hip, profiles/): a wave whose 64 lanes each fetch their own 64 B node
// costs the CU's vector-memory pipeline ~340 cycles per step whatever the occupancy -- per-lane walks are bound by
// that pipeline, not by arithmetic or HBM.  A packet walks the UNION of its rays' subtrees instead: the node index is
// wave-uniform, so nodes, leaf entries and triangle records come through the scalar cache (s_load) and the vector
// memory pipeline is not used at all; each lane still applies the reference's tests to its own ray, and only to
// nodes whose parent box its own ray passed (the lane mask travels with every stack entry), so every lane tests
// exactly the primitives its own per-lane walk would test.  The stack is one entry per WAVE, not per lane.
// ---------------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#if RT_PACKET_SCALAR_LOADS
typedef __attribute__((address_space(4))) const v4f cfloat4;   // constant address space + wave-uniform address => s_load
typedef __attribute__((address_space(4))) const v2f cfloat2;
typedef __attribute__((address_space(4))) const uint32_t cuint32;
#else
// Same-address vector loads (one cache line per wave instruction, served by the CU's L1).  The scalar cache is a
// few KB shared between CUs: a 4 MB tree walked by ~80 waves per scalar cache misses on almost every node and the
// scalar miss path then costs ~14k cycles per step (measured), against ~1-2k for a broadcast vector load.
typedef const v4f cfloat4;
typedef const v2f cfloat2;
typedef const uint32_t cuint32;
#endif

__device__ __forceinline__ NodeRegs sloadNode(const rt_bvh_node* nodes, int32_t node) {   // node must be wave-uniform
	cfloat4* q = (cfloat4*)(uintptr_t)(nodes + node);
	const v4f a = q[0], b = q[1], c = q[2];
	const v2f d = *(cfloat2*)(uintptr_t)(reinterpret_cast<const float4*>(nodes + node) + 3);
	NodeRegs r;
	r.b0[0] = a.x; r.b0[1] = a.y; r.b0[2] = a.z; r.b0[3] = a.w; r.b0[4] = b.x; r.b0[5] = b.y;
	r.b1[0] = b.z; r.b1[1] = b.w; r.b1[2] = c.x; r.b1[3] = c.y; r.b1[4] = c.z; r.b1[5] = c.w;
	r.c0 = __float_as_int(d.x); r.c1 = __float_as_int(d.y);
	return r;
}

// object->rayhit(e, d) with a wave-uniform primitive reference: the record is fetched once per wave (scalar loads)
template <bool CSG>
__device__ __forceinline__ float primHitUniform(const DevScene& S, uint32_t ref, f3 e, f3 d, uint32_t& tag) {
	const uint32_t ix = RT_PRIM_INDEX(ref);
	const uint32_t ty = RT_PRIM_TYPE(ref);
	if (ty == RT_PRIM_TRI) {
		cfloat4* q = (cfloat4*)(uintptr_t)(S.tris + ix);
		const v4f a = q[0], b = q[1], c = q[2];
		return triHit(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), mk3(c.y, c.z, c.w), e, d, false);
	}
	if (ty == RT_PRIM_SPHERE) {
		const v4f s = *(cfloat4*)(uintptr_t)(S.spheres + ix);
		return sphereHit(mk3(s.x, s.y, s.z), s.w, e, d, false);
	}
	if (ty == RT_PRIM_BOX) {
		const v4f a = *(cfloat4*)(uintptr_t)(S.boxes + ix);
		const v2f b = *(cfloat2*)(uintptr_t)(S.boxes[ix].b + 4);
		const float bb[6] = { a.x, a.y, a.z, a.w, b.x, b.y };
		return boxHit(bb, e, d, false);
	}
	if (CSG) return csgEval(S, S.csgs[ix], e, d, false, tag);
	return 0.0f;
}

// Treelet staging.  A packet's walk is a chain of dependent fetches (child addresses are only known once the parent
// has arrived) and each fetch from L2 costs a few thousand cycles, so the chain length, not bandwidth or arithmetic,
// set the time of the first packet kernels.  The tree is therefore stored a second time as 1 KB treelets of up to 15
// nodes, >= 4 levels each (rt_kernels.h): entering a treelet, the 64 lanes fetch it with ONE 16 B-per-lane load
// and park it in the wave's LDS; the nodes inside are then read from LDS (all lanes the same address: a broadcast),
// and only crossing into the next treelet touches memory again -- a 16-level tree is 4 dependent fetches deep instead
// of 16.  One LDS buffer per treelet LEVEL: a depth-first walk only ever returns to ancestors of the treelet it is
// in, and those are still resident in their level's buffer, so every treelet is fetched once per packet.
__device__ __forceinline__ uint32_t waveNumber() { return (blockIdx.x * kBlock + threadIdx.x) >> 6; }
__device__ __forceinline__ uint32_t waveCount() { return gridDim.x * (kBlock / 64); }

// RT_PROF builds: four waves of each packet kernel record (tag, value, shader clock) at every step of their walk
// into RtCounters::trace; tools/trace_summary.py turns that into per-phase cycle counts.  Tags: 1 packet start,
// 2 rays loaded + init done, 3 node step start (value = slot), 4 treelet staged, 5 node read + classified,
// 6 step end, 7 leaf start (value = count), 8 leaf end, 9 walk done, 10 next packet fetched, 11 results stored
#if RT_PROF
struct Tracer { unsigned long long* buf; uint32_t n; bool on; uint32_t hist[8]; };
__device__ __forceinline__ Tracer tracerFor(RtCounters* ctr, int kernel) {
	Tracer tr;
	const uint32_t w = waveNumber();
	const int slot = w == 100u ? 0 : w == 1000u ? 1 : w == 2000u ? 2 : w == 3000u ? 3 : -1;
	tr.on = slot >= 0; tr.n = 0; tr.buf = ctr->trace[kernel][slot < 0 ? 0 : slot];
	for (int i = 0; i < 8; i++) tr.hist[i] = 0;
	return tr;
}
#define TR_DECL(kernel) Tracer tr = tracerFor(ctr, kernel); Tracer* trp = &tr; \
	const uint32_t wlK = kernel; const uint32_t wlStart = (uint32_t)wall_clock64(); uint32_t wlLast = wlStart, wlMax = 0, wlN = 0, wlSteps = 0, wlLaneMax = 0, wlLaneSum = 0, wlLanes = 0;
#define WL_PACKET(T, alive) do { const uint32_t _n = (uint32_t)wall_clock64(); if (_n - wlLast > wlMax) { wlMax = _n - wlLast; \
		uint32_t _c = (alive) ? (uint32_t)(T).cntNodes : 0u, _m = _c, _s = _c; \
		for (int _o = 32; _o; _o >>= 1) { const uint32_t _a = __shfl_xor(_m, _o), _b = __shfl_xor(_s, _o); _m = _a > _m ? _a : _m; _s += _b; } \
		wlSteps = __builtin_amdgcn_readfirstlane((T).steps); wlLaneMax = _m; wlLaneSum = _s; wlLanes = (uint32_t)__popcll(__ballot(alive)); } wlLast = _n; wlN++; } while (0)
#define WL_END() do { if (__lane_id() == 0) for (int _i = 0; _i < 7; _i++) atomicAdd(&ctr->laneHist[wlK][_i], (unsigned long long)tr.hist[_i]); if (__lane_id() == 0 && waveNumber() < 8192u) { uint32_t* _w = ctr->waveLog[wlK][waveNumber()]; _w[0] = wlStart; _w[1] = (uint32_t)wall_clock64(); _w[2] = wlMax; _w[3] = wlN; \
		_w[4] = wlSteps; _w[5] = wlLaneMax; _w[6] = wlLaneSum; _w[7] = wlLanes; } } while (0)
#define TR(tag, val) do { if (trp->on) { const unsigned long long _t = __builtin_readcyclecounter(); \
	if (__lane_id() == 0 && trp->n < RT_TRACE_LEN) trp->buf[trp->n] = ((unsigned long long)(tag) << 56) | ((unsigned long long)((val) & 0xffffu) << 40) | (_t & 0xffffffffffull); trp->n++; } } while (0)
#define TR_ARG , trp
#else
struct Tracer;
#define TR_DECL(kernel)
#define WL_PACKET(T, alive) do { } while (0)
#define WL_END() do { } while (0)
#define TR(tag, val) do { } while (0)
#define TR_ARG
#endif

struct PacketLds {
	float4* root;               // [64] the root treelet, shared by the workgroup's waves
	float4* treelet;            // [levels - 1 (+ 1)][64]: this wave's buffer per deeper treelet level, then the staging buffer
	uint32_t (*stack)[8];       // [RT_PACKET_STACK_DEPTH] x 32 B: node, lane mask lo/hi, level (inner) or aux (leaf); the two-ray walk uses all 32 B
	uint32_t staging;           // index of the staging buffer in treelet[]
	float* cold;                // two-ray shadow walk only: [RT_COLD_FLOATS][64] per-lane state that only leaves touch (this lane's column)
	__device__ __forceinline__ float4* level(uint32_t l) const { return l == 0 ? root : treelet + (l - 1u) * (RT_QUAD_SLOTS * 4u); }
};
// Per-lane state of the two-ray shadow walk that only the leaves (and the rare exact box test) need -- the two directions, their
// lengths, the two attenuation colours -- lives in LDS, one column per lane, and is loaded where it is used: 14 registers fewer are
// live through the node steps, which is what decides how many waves a SIMD holds.  cold[k * 64] = value k of this lane.
enum { COLD_DA = 0, COLD_DB = 3, COLD_LENA = 6, COLD_LENB = 7, COLD_SHA = 8, COLD_SHB = 11 };
__device__ __forceinline__ f3 coldLd3(const float* cold, int k) { return mk3(cold[k * 64], cold[(k + 1) * 64], cold[(k + 2) * 64]); }
__device__ __forceinline__ void coldSt3(float* cold, int k, f3 v) { cold[k * 64] = v.x; cold[(k + 1) * 64] = v.y; cold[(k + 2) * 64] = v.z; }
extern __shared__ float4 packetLdsRaw[];
// Carves up the workgroup's dynamic LDS (RT_PACKET_LDS_BYTES) and fetches the root treelet.  Ends with a barrier:
// call it once, at the top of the kernel, from every thread.
__device__ __forceinline__ PacketLds packetLdsInit(const DevScene& S, bool withCold = false) {
	PacketLds L;
	const uint32_t perWave = (RT_PACKET_WAVE_LDS_BYTES(S.quad_levels, S.treelet_staging) + (withCold ? RT_PACKET_COLD_BYTES : 0)) / 16;
	L.root = packetLdsRaw;
	L.treelet = packetLdsRaw + RT_QUAD_SLOTS * 4 + (threadIdx.x >> 6) * perWave;
	L.staging = S.quad_levels > 1 ? (uint32_t)S.quad_levels - 1u : 0u;
	L.stack = reinterpret_cast<uint32_t (*)[8]>(L.treelet + (L.staging + (S.treelet_staging ? 1u : 0u)) * (RT_QUAD_SLOTS * 4u));
	L.cold = reinterpret_cast<float*>(L.stack + RT_PACKET_STACK_DEPTH) + (threadIdx.x & 63u);
	if (threadIdx.x < RT_QUAD_SLOTS * 4 && S.treelet_root >= 0) L.root[threadIdx.x] = S.quads[threadIdx.x];
	__syncthreads();
	return L;
}

// One generic primitive (wave-uniform reference, record fetched from memory) for the lanes in `inSub`.
template <bool ANYHIT, bool CSG, bool COUNT>
__device__ __forceinline__ void packetPrim(const DevScene& S, Trav& T, bool inSub, uint32_t ref, uint32_t key, float* coldShadow = nullptr) {
	if (inSub && (!ANYHIT || T.lit)) {
		uint32_t tag = 0;
		const float t = primHitUniform<CSG>(S, ref, T.e, T.d, tag);
		if (COUNT) T.cntPrims++;
		if (!ANYHIT) {   // bvh.cpp:109-118
			if ((double)t > 1e-5 && (t < T.tmin || (t == T.tmin && key < T.bestKey))) { T.best = ref; T.bestTag = tag; T.tmin = t; T.bestKey = key; }
		} else if (t < 1.0f && (double)(t * T.len) > 1e-5) {   // bvh.cpp:145-157
			const float* kt = S.materials[primMaterial(S, ref)].transmissive;
			if (!isZero3(kt)) { if (coldShadow) coldSt3(coldShadow, 0, coldLd3(coldShadow, 0) * ld3(kt)); else T.shadow = T.shadow * ld3(kt); }
			else T.lit = false;
		}
	}
}

// bvh.cpp:109-118 / 145-157 for a triangle whose record is already in registers (same in every lane)
template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ void packetTri(const DevScene& S, Trav& T, bool inSub, float4 a, float4 b, float4 c, uint32_t index, int mat, uint32_t key, float* coldShadow = nullptr) {
	const bool active = inSub && (!ANYHIT || T.lit);
	if (COUNT && active) T.cntPrims++;
#if RT_ABLATE & 1
	if (__float_as_uint(a.x) != 0x12345678u) return;   // diagnostic builds only: what the walks cost without their triangle tests
#endif
	// Most tests end at the plane: a shadow ray leaving a surface looks at the back of the neighbouring triangles.  The
	// signs of the plane test's two dot products say so before the fp64 divide and the three edge tests, and when no lane
	// of the packet is left the wave skips them (exact: triCandidate, rt_device.h).
	bool cand = active;
	if (ANYHIT) {   // (camera rays mostly face what they test: nothing to gain there)
		cand = active && triCandidate<ANYHIT>(mk3(a.x, a.y, a.z), mk3(c.y, c.z, c.w), T.e, T.d);
		if (__ballot(cand) == 0) return;
	}
	if (cand) {
		const float t = triHit(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), mk3(c.y, c.z, c.w), T.e, T.d, false);
		if (!ANYHIT) {
			if ((double)t > 1e-5 && (t < T.tmin || (t == T.tmin && key < T.bestKey))) { T.best = RT_PRIM_REF(RT_PRIM_TRI, index); T.bestTag = 0; T.tmin = t; T.bestKey = key; }
		} else if (t < 1.0f && (double)(t * T.len) > 1e-5) {
			const float* kt = S.materials[mat].transmissive;
			if (!isZero3(kt)) { if (coldShadow) coldSt3(coldShadow, 0, coldLd3(coldShadow, 0) * ld3(kt)); else T.shadow = T.shadow * ld3(kt); }
			else T.lit = false;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Quad nodes.  The packet walks read the tree in a second form (DevScene::quads, built at upload beside the binary treelets the
// per-lane walk keeps): a node holds the boxes of up to FOUR children -- the grandchildren of the binary node it stands for, or a
// child itself where that is a leaf or roots a sub-tree of another regime (exact reference boxes above, conservative ones below a fat
// leaf: a box is only skipped between boxes of one regime).  A step of the packet walk costs ~100 instructions before it has looked
// at a box (node fetch, ballots, masks, stack, branches) and the kernels are bound by instruction issue: one step for two levels
// pays that once, and the two child boxes in between are not tested at all.  Skipping a box is exact: the slab arithmetic is
// monotone in the bounds, a box that contains another is entered no later and left no earlier in floating point as well, and the
// reference's boxes nest exactly (min / max over subsets of the same primitives), so a ray that passes the inner box with the
// reference's arithmetic would have passed every box around it.  Layout (rt_kernels.h): 128 B = two 64 B slots of a 1 KB treelet.
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// Box verdicts as lane masks (round 3).  Which rays of the packet are in the current node is a 64-bit mask in scalar registers,
// not a flag per lane, and a box test yields two masks straight out of its two compares:
//   H  the ray certainly passes the box within its limit,      N  it is not certain that it misses
// so that "passed and in the node" is one s_and_b64 for 64 rays, and the stack, the pops and the choice of the next entry never touch
// a vector register.  Per lane the test is  lo = max(tnear, 0),  u = min(tfar, limit),  visit <=> lo <= u : with the ray's start
// folded into lo that is exactly the reference's  "tnear <= tfar, tfar >= 0, entry <= limit"  (entry = max(tnear, 0); its +-MAX_T
// clamps are subsumed, every limit being <= MAX_T), and the zero and the limit ride in the min3 / max3 that combine the slabs.  The
// multiply by the ray's 1/d is within 3 ulp of the reference's divide (boxClassify above), so  u - lo > m  with
// m = 1e-6 (lo + |u|) + 1e-30  (8 ulp) is the reference's "visit" and  u - lo < -m  its "skip"; between the two (N and not H) nothing
// is decided.  Lanes whose ray is NOT in the node get the limit -1: u <= -1 < 0 <= lo, never H; N is masked with the node's mask.
//
// Where "undecided" needs the reference's own arithmetic, and where not (round 3).  What the reference returns depends on the
// verdicts of its LEAF boxes only: its boxes nest exactly and the slab arithmetic is monotone in the bounds (see "Quad nodes" above),
// so a ray that passes a leaf's box has passed every box around it, and the set of leaves the reference visits is the set of leaves
// whose own box it passes.  An inner entry may therefore be visited on N alone -- a superset costs a step, never a result -- and the
// exact divide-based test (boxIntersect) runs only for undecided lanes of entries whose verdict is final: leaves of the reference's
// tree, and entries that root the conservative sub-tree under a fat leaf (their box is that leaf's box).  Inside such sub-trees every
// box is conservative anyway (undecided = visit).  Rays with a non-finite 1/d (`special`: 0 * inf = NaN, which v_min / v_max drop
// silently) cannot trust H or N: in the reference's tree they take the exact test at every entry, in a sub-tree they visit.
// ---------------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
__device__ __forceinline__ bool laneOf(u64 mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }   // this lane's bit, as a condition code (no vector instruction)

struct QuadHead { int32_t c[4]; uint32_t x[4]; };   // child references and aux words of a quad (wave-uniform)
__device__ __forceinline__ QuadHead quadHead(const float4* q) {
	const float4 q6 = q[6], q7 = q[7];
	QuadHead h;
	h.c[0] = __builtin_amdgcn_readfirstlane(__float_as_int(q6.x)); h.c[1] = __builtin_amdgcn_readfirstlane(__float_as_int(q6.y));
	h.c[2] = __builtin_amdgcn_readfirstlane(__float_as_int(q6.z)); h.c[3] = __builtin_amdgcn_readfirstlane(__float_as_int(q6.w));
	h.x[0] = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(q7.x)); h.x[1] = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(q7.y));
	h.x[2] = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(q7.z)); h.x[3] = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(q7.w));
	return h;
}
// is the verdict on this entry's box final (see above)?  `cons`: the quad itself belongs to a conservative sub-tree
__device__ __forceinline__ bool entryIsFinal(int32_t c, bool cons) { return !cons && (c < 0 || ((uint32_t)c & RT_TREELET_CONS) != 0u); }

// the slab differences of a pair of boxes (three 16 B words of a quad) from an origin: shared by every ray that starts there
struct PairDiff { v2f_ x0, y0, z0, x1, y1, z1; };
__device__ __forceinline__ PairDiff pairDiff(const float4& a, const float4& b, const float4& c, const f3& e) {
	PairDiff D;
	D.x0 = (v2f_){ a.x, a.y } - e.x; D.y0 = (v2f_){ a.z, a.w } - e.y; D.z0 = (v2f_){ b.x, b.y } - e.z;
	D.x1 = (v2f_){ b.z, b.w } - e.x; D.y1 = (v2f_){ c.x, c.y } - e.y; D.z1 = (v2f_){ c.z, c.w } - e.z;
	return D;
}
// verdict masks of the two boxes for one ray (direction 1/d = inv, limit tl: -1 in lanes that are not in the node); lo = max(tnear, 0)
__device__ __forceinline__ void pairVerdict(const PairDiff& D, const f3& inv, float tl, u64& H0, u64& N0, u64& H1, u64& N1, float& lo0, float& lo1) {
	const v2f_ tx0 = D.x0 * inv.x, ty0 = D.y0 * inv.y, tz0 = D.z0 * inv.z;
	const v2f_ tx1 = D.x1 * inv.x, ty1 = D.y1 * inv.y, tz1 = D.z1 * inv.z;
	lo0 = fmaxf(fmaxf(fminf(tx0.x, tx0.y), fminf(ty0.x, ty0.y)), fmaxf(fminf(tz0.x, tz0.y), 0.0f));
	lo1 = fmaxf(fmaxf(fminf(tx1.x, tx1.y), fminf(ty1.x, ty1.y)), fmaxf(fminf(tz1.x, tz1.y), 0.0f));
	const float u0 = fminf(fminf(fmaxf(tx0.x, tx0.y), fmaxf(ty0.x, ty0.y)), fminf(fmaxf(tz0.x, tz0.y), tl));
	const float u1 = fminf(fminf(fmaxf(tx1.x, tx1.y), fmaxf(ty1.x, ty1.y)), fminf(fmaxf(tz1.x, tz1.y), tl));
	const v2f_ lo = { lo0, lo1 }, u = { u0, u1 };
	const v2f_ g = u - lo;
	const v2f_ au = { fabsf(u0), fabsf(u1) };
	// (the margin itself need not match anything bit for bit: a fused multiply-add, packed for the two boxes)
	const v2f_ m = __builtin_elementwise_fma((v2f_){ 1e-6f, 1e-6f }, lo + au, (v2f_){ 1e-30f, 1e-30f });
	H0 = __ballot(g.x > m.x); N0 = __ballot(!(g.x < -m.x));
	H1 = __ballot(g.y > m.y); N1 = __ballot(!(g.y < -m.y));
}
// the reference's own test (objects.cpp:11-44 as bvh.cpp:97-100 / 131-134 use it) for the lanes of `lanes`: the ones it lets through.
// NEAREST: the limit is the best hit so far and a box this close to it is visited whatever the comparison says (see travRound).
template <bool NEAREST>
__device__ __forceinline__ u64 boxExact(const float* b, const f3& e, const f3& d, u64 lanes, float tlimit) {
	bool ok = false;
	if (laneOf(lanes)) { const float t = boxIntersect(b, e, d, false); ok = !(t < 0 || (!NEAREST && t > tlimit)); }
	return __ballot(ok);
}

// `alive`: the lane holds a query.  T must come from travInit (T.cur == kEmpty: the lane's ray missed the root box).
// DEFER (shadow queries of scenes with at most RT_PACKET_STACK_DEPTH CSG objects): a CSG leaf is not evaluated where the walk meets it.
// The walk notes the object and the lanes whose rays reached it in the unused half of a stack slot (a leaf is met once per packet, so
// one slot per object is enough) and the objects are evaluated when the walk is over.  The evaluation's registers then do not overlap
// the walk's: with csgEval inside the loop the area-light kernel spilled in its node steps and C4 paid 1.3 of its 9.4 ms for having
// CSG objects at all, whether a ray came near them or not.  Any order is correct for a shadow query.
template <bool ANYHIT, bool CSG, bool COUNT, bool DEFER = false>
__device__ __forceinline__ void packetTraverse(const DevScene& S, const PacketLds& L, Trav& T, bool alive, Tracer* trp = nullptr) {
	static_assert(!DEFER || (ANYHIT && CSG), "deferred CSG leaves: shadow queries only");
	const uint32_t lane = __lane_id();
	u64 in = __ballot(alive && T.cur != kEmpty);   // the lanes whose ray passed the box of the current node
	if (in == 0) return;
	const u64 special = __ballot(alive && T.special);
	u64 lit = ~0ull;                               // ANYHIT: lanes whose ray has not met an opaque occluder yet
	int32_t node = S.treelet_root;
	uint32_t info = node < 0 ? S.treelet_root_aux : 0u;   // inner node: treelet level; leaf: aux
	int sp = 0;
	int nDeferred = 0;
	bool walking = true;
	// (Measured and dropped: raising the wave's issue priority with the length of its walk, and prefetching a pushed
	// treelet with an asynchronous global->LDS copy into a scratch buffer -- neither moved the launch time: the long
	// walks that end a launch are bound by their own chain of dependent steps.  Their tails are instead filled with
	// other shards of the frame, see rt_capi_device.hip.)
	while (walking) {
		node = __builtin_amdgcn_readfirstlane(node);
		info = __builtin_amdgcn_readfirstlane(info);
		bool popNext = true;
		if (node >= 0) {
			TR(3, node & 15);
#if RT_PROF
			T.steps++;
			{ const int act = __popcll(in); const int b = act <= 2 ? 0 : act <= 4 ? 1 : act <= 8 ? 2 : act <= 16 ? 3 : act <= 32 ? 4 : act <= 48 ? 5 : 6; if (trp) trp->hist[b]++; }
#endif
			const bool cons = ((uint32_t)node & RT_TREELET_CONS) != 0u;   // a node of a sub-tree under a fat leaf: conservative box tests
			const int32_t nidx = (int32_t)((uint32_t)node & (RT_TREELET_CONS - 1u));
			float4* buf = L.level(info);
			if ((nidx & (RT_QUAD_SLOTS - 1)) == 0 && nidx != 0) {   // the root of a deeper treelet: stage it
#pragma unroll
#if RT_ABLATE & 2
				// diagnostic builds only: every treelet fetched twice, the second fetch depending on the first (what do the fetches' round trips cost?)
				for (uint32_t k = 0; k < RT_QUAD_SLOTS * 4u; k += 64u) {
					const float4 once = S.quads[(size_t)nidx * 4 + k + lane];
					uint32_t z = __float_as_uint(once.x);
					asm volatile("v_and_b32 %0, 0, %0" : "+v"(z));
					buf[k + lane] = S.quads[(size_t)nidx * 4 + k + lane + z];
				}
#else
				for (uint32_t k = 0; k < RT_QUAD_SLOTS * 4u; k += 64u) buf[k + lane] = S.quads[(size_t)nidx * 4 + k + lane];
#endif
				__builtin_amdgcn_wave_barrier();
				TR(4, nidx >> 4);
			}
			TR(12, 0);
			const float4* q = buf + (nidx & (RT_QUAD_SLOTS - 1)) * 4;
			const QuadHead Q = quadHead(q);
			const float tlimit = ANYHIT ? 1.0f : T.tmin;
			const float tl = laneOf(in) ? tlimit : -1.0f;
			const uint32_t firstLane = (uint32_t)__ffsll((long long)in) - 1u;
			const u64 spIn = special & in;
			u64 m[4];
			uint32_t key[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
			if (COUNT || RT_PROF) T.cntNodes += laneOf(in) ? 1u : 0u;
			if (COUNT) T.cntSteps++;
#pragma unroll
			for (int pair = 0; pair < 2; pair++) {   // two boxes at a time: three 16 B words hold a pair
				if (pair == 1 && Q.c[2] == RT_QUAD_EMPTY) { m[2] = 0; m[3] = 0; continue; }   // a quad of two entries (entries fill from the front)
				const float4 a = q[3 * pair], b = q[3 * pair + 1], c = q[3 * pair + 2];
				const PairDiff D = pairDiff(a, b, c, T.e);
				u64 H0, N0, H1, N1;
				float lo0, lo1;
				pairVerdict(D, T.inv, tl, H0, N0, H1, N1, lo0, lo1);
				const u64 valid1 = (pair == 1 && Q.c[3] == RT_QUAD_EMPTY) ? 0ull : ~0ull;   // (wave-uniform)
				u64 v0 = N0 & in, v1 = N1 & in & valid1;
				if (cons) { v0 |= spIn; v1 |= spIn & valid1; }   // too close to call = visit
				else {
					const u64 a0 = (entryIsFinal(Q.c[2 * pair], false) ? v0 & ~H0 : 0ull) | spIn;
					const u64 a1 = ((entryIsFinal(Q.c[2 * pair + 1], false) ? v1 & ~H1 : 0ull) | spIn) & valid1;
					if ((a0 | a1) != 0) {   // too close to call where it counts (rare): the reference's own arithmetic decides
						const float b0[6] = { a.x, a.y, a.z, a.w, b.x, b.y };
						const float b1[6] = { b.z, b.w, c.x, c.y, c.z, c.w };
						if (a0 != 0) v0 = (v0 & ~a0) | boxExact<!ANYHIT>(b0, T.e, T.d, a0, tlimit);
						if (a1 != 0) v1 = (v1 & ~a1) | boxExact<!ANYHIT>(b1, T.e, T.d, a1, tlimit);
					}
				}
				m[2 * pair] = v0; m[2 * pair + 1] = v1;
				if (!ANYHIT) {   // visiting keys: the entry distance of the wave's first ray in the node (any order is correct; near first prunes more)
					if (v0 != 0) key[2 * pair] = ((uint32_t)__builtin_amdgcn_readlane(__float_as_int(lo0), (int)firstLane) & ~3u) | (uint32_t)(2 * pair);
					if (v1 != 0) key[2 * pair + 1] = ((uint32_t)__builtin_amdgcn_readlane(__float_as_int(lo1), (int)firstLane) & ~3u) | (uint32_t)(2 * pair + 1);
				}
			}
			TR(5, __popcll(m[0] | m[1] | m[2] | m[3]));
			// The entry to go on with: the nearest one for the wave's first ray (nearest-hit walk), the first one any ray passed (shadow
			// rays: any order is correct).  The other entries that some ray passed wait on the stack.  Written out per case: the entry's
			// words are in scalar registers, and picking them by a run-time index costs more instructions than the step's box tests.
			// (sp < RT_PACKET_STACK_DEPTH always holds: rt_scene_upload refuses a tree that does not pass RT_PACKET_STACK_FITS, rt_kernels.h)
			// (sp <= RT_PACKET_STACK_DEPTH - 3 holds here, see above; one clamp per step keeps a broken invariant inside the stack's memory)
			sp = sp > RT_PACKET_STACK_DEPTH - 3 ? RT_PACKET_STACK_DEPTH - 3 : sp;
#define RT_Q_INF(i) Q.x[i]   /* leaf: its aux word; inner entry: the treelet level of its quad */
#define RT_Q_PUSH(i) do { if (m[i] != 0) { *reinterpret_cast<uint4*>(L.stack[sp]) = make_uint4((uint32_t)Q.c[i], (uint32_t)m[i], (uint32_t)(m[i] >> 32), RT_Q_INF(i)); sp++; } } while (0)
#define RT_Q_TAKE(i) do { node = Q.c[i]; info = RT_Q_INF(i); in = m[i]; popNext = false; } while (0)
			uint32_t first;
			if (!ANYHIT) {
				const uint32_t k01 = key[0] < key[1] ? key[0] : key[1], k23 = key[2] < key[3] ? key[2] : key[3];
				const uint32_t kmin = k01 < k23 ? k01 : k23;
				first = kmin == 0xffffffffu ? 4u : kmin & 3u;
			} else first = m[0] ? 0u : m[1] ? 1u : m[2] ? 2u : m[3] ? 3u : 4u;
			if (first == 0u) { RT_Q_PUSH(3); RT_Q_PUSH(2); RT_Q_PUSH(1); RT_Q_TAKE(0); }
			else if (first == 1u) { RT_Q_PUSH(3); RT_Q_PUSH(2); if (!ANYHIT) RT_Q_PUSH(0); RT_Q_TAKE(1); }
			else if (first == 2u) { RT_Q_PUSH(3); if (!ANYHIT) { RT_Q_PUSH(1); RT_Q_PUSH(0); } RT_Q_TAKE(2); }
			else if (first == 3u) { if (!ANYHIT) { RT_Q_PUSH(2); RT_Q_PUSH(1); RT_Q_PUSH(0); } RT_Q_TAKE(3); }
#undef RT_Q_INF
#undef RT_Q_PUSH
#undef RT_Q_TAKE
			TR(6, sp);
		} else if (info != RT_TREELET_NO_RUN) {   // leaf made of consecutive triangles: no leaf_prims indirection
			const bool inSub = laneOf(in);
			const bool longRun = (info >> 26) == 0u;   // external run in its long form: info = count, first triangle = first leaf entry
			const uint32_t count = longRun ? info : info >> 26;
			TR(7, count);
			const uint32_t key0 = (uint32_t)~node + 1u;
			if (!longRun && (info & RT_TREELET_EMBED)) {   // ... which arrived with the treelet
				const float4* q = L.level((info >> RT_QUAD_SLOT_BITS) & 15u) + (info & (RT_QUAD_SLOTS - 1u)) * 4u;
				for (uint32_t k = 0; k < count; k++, q += 4) {
					const float4 a = q[0], b = q[1], c = q[2], d = q[3];
					packetTri<ANYHIT, COUNT>(S, T, inSub, a, b, c, __float_as_uint(d.z), __float_as_int(d.y), __float_as_uint(d.w));   // slot = triangle, material, index, key
				}
			} else {   // ... fetched 16 at a time (64 lanes x 16 B) into the staging buffer
				const uint32_t first = longRun ? (uint32_t)~node : info & (RT_TREELET_EMBED - 1u);
				float4* buf = L.treelet + L.staging * (RT_QUAD_SLOTS * 4u);
				for (uint32_t k0 = 0; k0 < count; k0 += 16u) {
					const uint32_t nb = count - k0 < 16u ? count - k0 : 16u;
					if (lane < nb * 4u) buf[lane] = reinterpret_cast<const float4*>(S.tris + first + k0)[lane];
					__builtin_amdgcn_wave_barrier();
					for (uint32_t k = 0; k < nb; k++) {
						const float4 a = buf[4 * k], b = buf[4 * k + 1], c = buf[4 * k + 2], d = buf[4 * k + 3];
						packetTri<ANYHIT, COUNT>(S, T, inSub, a, b, c, first + k0 + k, __float_as_int(d.y), key0 + k0 + k);
					}
					__builtin_amdgcn_wave_barrier();
				}
			}
			if (ANYHIT) lit = __ballot(T.lit);
			TR(8, 0);
		} else {   // leaf: bvh.cpp:109-118 / 145-157 for the lanes whose ray passed this leaf's box
			const bool inSub = laneOf(in);
			int32_t at = ~node;
			for (;;) {
				const uint32_t entry = *(cuint32*)(uintptr_t)(S.leaf_prims + at);
				if (DEFER && RT_PRIM_TYPE(entry & ~RT_PRIM_LAST) == RT_PRIM_CSG) {   // (wave-uniform)
					const u64 md = in & __ballot(T.lit);
					if (md != 0 && nDeferred < RT_PACKET_STACK_DEPTH) {   // (always: DEFER variants are launched only when RT_PACKET_DEFER_FITS, rtk_shadow_area)
						if (lane == 0) reinterpret_cast<uint4*>(L.stack[nDeferred])[1] = make_uint4(entry & ~RT_PRIM_LAST, (uint32_t)md, (uint32_t)(md >> 32), (uint32_t)at + 1u);
						nDeferred++;
					}
				} else packetPrim<ANYHIT, CSG && !DEFER, COUNT>(S, T, inSub, entry & ~RT_PRIM_LAST, (uint32_t)at + 1u);
				if (entry & RT_PRIM_LAST) break;
				at++;
			}
			if (ANYHIT) lit = __ballot(T.lit);
		}
		if (popNext) {
			for (;;) {
				if (sp == 0) { walking = false; break; }
				sp--;
				const uint4 entry = *reinterpret_cast<const uint4*>(L.stack[sp]);
				node = __builtin_amdgcn_readfirstlane((int)entry.x);
				info = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry.w);
				in = (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)entry.y) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)entry.z) << 32);
				if (ANYHIT) in &= lit;
				if (in != 0) break;   // nobody left who wants this subtree (occluded lanes)
			}
		}
	}
	if (DEFER) {
		for (int i = 0; i < nDeferred; i++) {
			const uint4 en = reinterpret_cast<const uint4*>(L.stack[i])[1];
			const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.x);
			const u64 md = (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)en.y) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)en.z) << 32);
			const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.w);
			const bool inD = laneOf(md) && T.lit;
			if (__ballot(inD) == 0) continue;
			packetPrim<ANYHIT, CSG, COUNT>(S, T, inD, ref, key);
		}
	}
}

// Two shadow rays per lane: the rays from one surface point towards two lights.  Both start inside the same leaves, so
// the nodes whose boxes contain the point -- most of what a shadow ray out of a height field visits -- are visited by
// both, and a step's fixed cost (node fetch, stack, branches) is paid once for the two, as are the slab differences (same
// origin).  Each ray keeps its own lane masks; every lane still applies the reference's tests to exactly the
// primitives its own walk of that ray would reach.  Any-hit only (no visiting order to respect).  A and B hold the same
// origin.
template <bool CSG, bool COUNT>
__device__ __forceinline__ void packetTraverse2(const DevScene& S, const PacketLds& L, Trav& A, Trav& B, bool aliveA, bool aliveB) {
	const uint32_t lane = __lane_id();
	u64 inA = __ballot(aliveA && A.cur != kEmpty), inB = __ballot(aliveB && B.cur != kEmpty);
	if ((inA | inB) == 0) return;
	const u64 specialA = __ballot(aliveA && A.special), specialB = __ballot(aliveB && B.special);
	u64 litA = ~0ull, litB = ~0ull;
	int32_t node = S.treelet_root;
	uint32_t info = node < 0 ? S.treelet_root_aux : 0u;
	int sp = 0;
	uint4* stack = reinterpret_cast<uint4*>(L.stack);   // two uint4 per entry: node, info, -, - | mask A lo/hi, mask B lo/hi
	for (;;) {
		node = __builtin_amdgcn_readfirstlane(node);
		info = __builtin_amdgcn_readfirstlane(info);
		bool popNext = true;
		if (node >= 0) {
			const bool cons = ((uint32_t)node & RT_TREELET_CONS) != 0u;
			const int32_t nidx = (int32_t)((uint32_t)node & (RT_TREELET_CONS - 1u));
			float4* buf = L.level(info);
			if ((nidx & (RT_QUAD_SLOTS - 1)) == 0 && nidx != 0) {
#pragma unroll
				for (uint32_t k = 0; k < RT_QUAD_SLOTS * 4u; k += 64u) buf[k + lane] = S.quads[(size_t)nidx * 4 + k + lane];
				__builtin_amdgcn_wave_barrier();
			}
			const float4* q = buf + (nidx & (RT_QUAD_SLOTS - 1)) * 4;
			const QuadHead Q = quadHead(q);
			const float tlA = laneOf(inA) ? 1.0f : -1.0f, tlB = laneOf(inB) ? 1.0f : -1.0f;
			const u64 spA = specialA & inA, spB = specialB & inB;
			u64 mA[4], mB[4];
			if (COUNT) { A.cntNodes += laneOf(inA) ? 1u : 0u; B.cntNodes += laneOf(inB) ? 1u : 0u; A.cntSteps++; }
#pragma unroll
			for (int pair = 0; pair < 2; pair++) {
				if (pair == 1 && Q.c[2] == RT_QUAD_EMPTY) { mA[2] = mA[3] = mB[2] = mB[3] = 0; continue; }   // a quad of two entries: nothing in the second pair
				const float4 a = q[3 * pair], b = q[3 * pair + 1], cc = q[3 * pair + 2];
				const PairDiff D = pairDiff(a, b, cc, A.e);
				u64 H0A, N0A, H1A, N1A, H0B, N0B, H1B, N1B;
				float lo0, lo1;
				pairVerdict(D, A.inv, tlA, H0A, N0A, H1A, N1A, lo0, lo1);
				pairVerdict(D, B.inv, tlB, H0B, N0B, H1B, N1B, lo0, lo1);
				const u64 valid1 = (pair == 1 && Q.c[3] == RT_QUAD_EMPTY) ? 0ull : ~0ull;   // (wave-uniform)
				u64 v0A = N0A & inA, v1A = N1A & inA & valid1, v0B = N0B & inB, v1B = N1B & inB & valid1;
				if (cons) { v0A |= spA; v1A |= spA & valid1; v0B |= spB; v1B |= spB & valid1; }   // too close to call = visit
				else {
					const bool f0 = entryIsFinal(Q.c[2 * pair], false), f1 = entryIsFinal(Q.c[2 * pair + 1], false);
					const u64 a0A = (f0 ? v0A & ~H0A : 0ull) | spA, a0B = (f0 ? v0B & ~H0B : 0ull) | spB;
					const u64 a1A = ((f1 ? v1A & ~H1A : 0ull) | spA) & valid1, a1B = ((f1 ? v1B & ~H1B : 0ull) | spB) & valid1;
					if ((a0A | a0B | a1A | a1B) != 0) {   // too close to call where it counts (rare): the reference's own arithmetic decides
						const f3 dA = coldLd3(L.cold, COLD_DA), dB = coldLd3(L.cold, COLD_DB);
						const float b0[6] = { a.x, a.y, a.z, a.w, b.x, b.y };
						const float b1[6] = { b.z, b.w, cc.x, cc.y, cc.z, cc.w };
						if (a0A != 0) v0A = (v0A & ~a0A) | boxExact<false>(b0, A.e, dA, a0A, 1.0f);
						if (a1A != 0) v1A = (v1A & ~a1A) | boxExact<false>(b1, A.e, dA, a1A, 1.0f);
						if (a0B != 0) v0B = (v0B & ~a0B) | boxExact<false>(b0, A.e, dB, a0B, 1.0f);
						if (a1B != 0) v1B = (v1B & ~a1B) | boxExact<false>(b1, A.e, dB, a1B, 1.0f);
					}
				}
				mA[2 * pair] = v0A; mA[2 * pair + 1] = v1A; mB[2 * pair] = v0B; mB[2 * pair + 1] = v1B;
			}
			// the first entry any ray passed is walked next, the others wait on the stack (any order is correct for shadow rays); written
			// out per case, see packetTraverse
			sp = sp > RT_PACKET_STACK_DEPTH - 3 ? RT_PACKET_STACK_DEPTH - 3 : sp;   // (never: see packetTraverse)
#define RT_Q_INF(i) Q.x[i]
#define RT_Q_PUSH(i) do { if ((mA[i] | mB[i]) != 0) { stack[2 * sp] = make_uint4((uint32_t)Q.c[i], RT_Q_INF(i), 0u, 0u); \
				stack[2 * sp + 1] = make_uint4((uint32_t)mA[i], (uint32_t)(mA[i] >> 32), (uint32_t)mB[i], (uint32_t)(mB[i] >> 32)); sp++; } } while (0)
#define RT_Q_TAKE(i) do { node = Q.c[i]; info = RT_Q_INF(i); inA = mA[i]; inB = mB[i]; popNext = false; } while (0)
			if ((mA[0] | mB[0]) != 0) { RT_Q_PUSH(3); RT_Q_PUSH(2); RT_Q_PUSH(1); RT_Q_TAKE(0); }
			else if ((mA[1] | mB[1]) != 0) { RT_Q_PUSH(3); RT_Q_PUSH(2); RT_Q_TAKE(1); }
			else if ((mA[2] | mB[2]) != 0) { RT_Q_PUSH(3); RT_Q_TAKE(2); }
			else if ((mA[3] | mB[3]) != 0) { RT_Q_TAKE(3); }
#undef RT_Q_INF
#undef RT_Q_PUSH
#undef RT_Q_TAKE
		} else if (info != RT_TREELET_NO_RUN) {
			const bool la = laneOf(inA), lb = laneOf(inB);
			const bool longRun = (info >> 26) == 0u;
			const uint32_t count = longRun ? info : info >> 26;
			const uint32_t key0 = (uint32_t)~node + 1u;
			// what only leaves need comes back from LDS (and is dead again after the leaf)
			A.d = coldLd3(L.cold, COLD_DA); B.d = coldLd3(L.cold, COLD_DB); A.len = L.cold[COLD_LENA * 64]; B.len = L.cold[COLD_LENB * 64];
			if (!longRun && (info & RT_TREELET_EMBED)) {
				const float4* q = L.level((info >> RT_QUAD_SLOT_BITS) & 15u) + (info & (RT_QUAD_SLOTS - 1u)) * 4u;
				for (uint32_t k = 0; k < count; k++, q += 4) {
					const float4 a = q[0], b = q[1], c = q[2], d = q[3];
					packetTri<true, COUNT>(S, A, la, a, b, c, __float_as_uint(d.z), __float_as_int(d.y), __float_as_uint(d.w), L.cold + COLD_SHA * 64);
					packetTri<true, COUNT>(S, B, lb, a, b, c, __float_as_uint(d.z), __float_as_int(d.y), __float_as_uint(d.w), L.cold + COLD_SHB * 64);
				}
			} else {
				const uint32_t first = longRun ? (uint32_t)~node : info & (RT_TREELET_EMBED - 1u);
				float4* buf = L.treelet + L.staging * (RT_QUAD_SLOTS * 4u);
				for (uint32_t k0 = 0; k0 < count; k0 += 16u) {
					const uint32_t nb = count - k0 < 16u ? count - k0 : 16u;
					if (lane < nb * 4u) buf[lane] = reinterpret_cast<const float4*>(S.tris + first + k0)[lane];
					__builtin_amdgcn_wave_barrier();
					for (uint32_t k = 0; k < nb; k++) {
						const float4 a = buf[4 * k], b = buf[4 * k + 1], c = buf[4 * k + 2], d = buf[4 * k + 3];
						packetTri<true, COUNT>(S, A, la, a, b, c, first + k0 + k, __float_as_int(d.y), key0 + k0 + k, L.cold + COLD_SHA * 64);
						packetTri<true, COUNT>(S, B, lb, a, b, c, first + k0 + k, __float_as_int(d.y), key0 + k0 + k, L.cold + COLD_SHB * 64);
					}
					__builtin_amdgcn_wave_barrier();
				}
			}
			litA = __ballot(A.lit); litB = __ballot(B.lit);
		} else {
			const bool la = laneOf(inA), lb = laneOf(inB);
			int32_t at = ~node;
			A.d = coldLd3(L.cold, COLD_DA); B.d = coldLd3(L.cold, COLD_DB); A.len = L.cold[COLD_LENA * 64]; B.len = L.cold[COLD_LENB * 64];
			for (;;) {
				const uint32_t entry = *(cuint32*)(uintptr_t)(S.leaf_prims + at);
				packetPrim<true, CSG, COUNT>(S, A, la, entry & ~RT_PRIM_LAST, (uint32_t)at + 1u, L.cold + COLD_SHA * 64);
				packetPrim<true, CSG, COUNT>(S, B, lb, entry & ~RT_PRIM_LAST, (uint32_t)at + 1u, L.cold + COLD_SHB * 64);
				if (entry & RT_PRIM_LAST) break;
				at++;
			}
			litA = __ballot(A.lit); litB = __ballot(B.lit);
		}
		if (popNext) {
			for (;;) {
				if (sp == 0) return;
				sp--;
				const uint4 e0 = stack[2 * sp], e1 = stack[2 * sp + 1];
				node = __builtin_amdgcn_readfirstlane((int)e0.x);
				info = (uint32_t)__builtin_amdgcn_readfirstlane((int)e0.y);
				inA = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)e1.x) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)e1.y) << 32)) & litA;
				inB = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)e1.z) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)e1.w) << 32)) & litB;
				if ((inA | inB) != 0) break;
			}
		}
	}
}

// Work distribution for the persistent kernels.  A single device-wide cursor saturates at roughly 90 returning
// atomics per microsecond (MI355X_MICROARCH.md, row "dequeue"), and one atomic per lane refill made that cursor the
// bottleneck of the whole launch.  Each wave therefore takes kWorkChunk consecutive indices per atomic into a
// wave-local pool and refills its idle lanes from the pool.
constexpr uint32_t kWorkChunk = 128;
constexpr uint32_t kPacketGroup = 8;
#ifndef RT_PACKET_CHUNK
#define RT_PACKET_CHUNK 1
#endif
constexpr uint32_t kPacketChunk = RT_PACKET_CHUNK;   // packets per cursor atomic (divides kPacketGroup); 2 and 4 measured slower: the statically handed-out share grows with it and unbalances the launch
struct WorkPool { uint32_t next, end, chunk; bool drained; };

__device__ __forceinline__ uint32_t fetchWork(bool need, WorkPool& pool, uint32_t* head, unsigned long long total) {
	const unsigned long long mask = __ballot(need);
	if (mask == 0) return kInactive;
	if (pool.next == pool.end && !pool.drained) {   // wave-uniform
		uint32_t base = 0;
		if (__lane_id() == 0) base = atomicAdd(head, pool.chunk);
		const unsigned long long b = (unsigned long long)__builtin_amdgcn_readfirstlane(base) + (unsigned long long)waveCount() * pool.chunk;
		if (b >= total) pool.drained = true;
		else {
			pool.next = (uint32_t)b;
			pool.end = (uint32_t)(b + pool.chunk < total ? b + pool.chunk : total);
		}
	}
	const uint32_t avail = pool.end - pool.next;
	const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << __lane_id()) - 1ull));
	const uint32_t mine = (need && rank < avail) ? pool.next + rank : kInactive;
	const uint32_t cnt = (uint32_t)__popcll(mask);
	pool.next += cnt < avail ? cnt : avail;
	return mine;
}
// every wave starts on the chunk with its own number (no cursor stampede at launch, see fetchPacket)
__device__ __forceinline__ WorkPool firstPool(unsigned long long total) {
	// a launch that does not fill the chip twice hands every wave one wave's worth: small levels of a deep ray tree end with
	// their slowest lane, and two rounds of 64 on half of the waves take twice as long as one round on all of them
	const uint32_t chunk = total <= (unsigned long long)waveCount() * 64u ? 64u : kWorkChunk;
	const unsigned long long b = (unsigned long long)waveNumber() * chunk;
	WorkPool p;
	p.chunk = chunk;
	// Nothing beyond the statically numbered pieces: no wave needs to ask the cursor to find that out.  (A launch sized by a bound, or
	// a small level of a deep ray tree, has thousands of waves and little work: each of them asking once is the stampede again.)
	p.drained = total <= (unsigned long long)waveCount() * chunk;
	p.next = (uint32_t)(b < total ? b : total);
	p.end = (uint32_t)(b + chunk < total ? b + chunk : total);
	if (p.end < p.next) p.end = p.next;
	return p;
}
__device__ __forceinline__ bool workLeft(const WorkPool& pool) { return !pool.drained || pool.next != pool.end; }
constexpr int kRefillBelow = RT_REFILL_BELOW;   // fetch new queries once fewer than this many of the 64 lanes are still busy

// =========================================================================================================
// kernels
// =========================================================================================================

__global__ __launch_bounds__(kBlock) void k_generate(rt_camera cam, int W, int H, int aaX, int aaY, int tileRank, int tileRanks,
                                                     uint32_t r0, uint32_t count, RtQueue rq) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= count) return;
	const uint32_t r = r0 + i, aaSq = (uint32_t)(aaX * aaY);
	const uint32_t lp = r / aaSq, k = r % aaSq;
	int x, y;
	if (!tilePixel(lp, W, H, tileRank, tileRanks, x, y)) { qsu(rq, RQ_PIX, i, kInactive); return; }
	// s(x,y) / s_aa(x,y,k), q1.cpp:51-64, generalised to an aaX x aaY sub-pixel grid (1x1 and 2x2 are the reference's)
	const float ox = ((float)(k % (uint32_t)aaX) + 0.5f) / (float)aaX, oy = ((float)(k / (uint32_t)aaX) + 0.5f) / (float)aaY;
	const f3 eye = ld3(cam.eye);
	const f3 s = eye + ld3(cam.facing) + ld3(cam.right) * (2 * (((float)x + ox) / (float)W - 0.5f)) + ld3(cam.up) * (2 * (((float)y + oy) / (float)H - 0.5f));
	const f3 d = s - eye;   // raytracer.cpp:316
	qsf(rq, RQ_OX, i, eye.x); qsf(rq, RQ_OY, i, eye.y); qsf(rq, RQ_OZ, i, eye.z);
	qsf(rq, RQ_DX, i, d.x); qsf(rq, RQ_DY, i, d.y); qsf(rq, RQ_DZ, i, d.z);
	qsf(rq, RQ_WR, i, 1.0f); qsf(rq, RQ_WG, i, 1.0f); qsf(rq, RQ_WB, i, 1.0f);
	qsu(rq, RQ_PIX, i, ((uint32_t)y * (uint32_t)W + (uint32_t)x) * aaSq + k);
	qsu(rq, RQ_PATH, i, 1u);
}

// rays given by the caller: origin + target (trace) or origin + direction (findNearest)
__global__ __launch_bounds__(kBlock) void k_load_rays(const float* a, const float* b, uint32_t n, int bIsTarget, RtQueue rq, uint32_t pixBase) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const f3 e = ld3(a + 3 * (size_t)i), s = ld3(b + 3 * (size_t)i);
	const f3 d = bIsTarget ? s - e : s;   // raytracer.cpp:316
	qsf(rq, RQ_OX, i, e.x); qsf(rq, RQ_OY, i, e.y); qsf(rq, RQ_OZ, i, e.z);
	qsf(rq, RQ_DX, i, d.x); qsf(rq, RQ_DY, i, d.y); qsf(rq, RQ_DZ, i, d.z);
	qsf(rq, RQ_WR, i, 1.0f); qsf(rq, RQ_WG, i, 1.0f); qsf(rq, RQ_WB, i, 1.0f);
	qsu(rq, RQ_PIX, i, pixBase + i);
	qsu(rq, RQ_PATH, i, 1u);
}

// A launch queued before the host knows how many items the level before it produced gets a bound for its grid (nArg) and reads
// the count itself (nDev: the previous level's append counter, capped by the bound, which is also the queue's capacity).
__device__ __forceinline__ uint32_t deviceCount(uint32_t nArg, const uint32_t* nDev) {
	if (!nDev) return nArg;
	const uint32_t n = *nDev;
	return n < nArg ? n : nArg;
}

// Persistent nearest-hit kernel: the grid is sized to fill the chip once; every wave keeps pulling rays (fetchWork)
// and lanes whose ray has finished are refilled while the rest of the wave keeps walking, so one long ray does not
// idle 63 lanes and the launch has no tail of half-empty waves.
template <bool CSG, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_intersect(DevScene S, RtQueue rq, uint32_t nArg, const uint32_t* nDev, RtQueue hq, RtCounters* ctr) {
	__shared__ int32_t stack[kStack][kBlock];
	const uint32_t n = deviceCount(nArg, nDev);
	uint32_t idx = kInactive;
	WorkPool pool = firstPool(n);
	Trav T;
	T.cur = kEmpty; T.sp = 0; T.cntNodes = 0; T.cntPrims = 0; T.cntSteps = 0;
	unsigned long long cn = 0, cp = 0;
	for (;;) {
		const uint32_t got = fetchWork(idx == kInactive, pool, &ctr->head_nearest, n);
		if (got != kInactive) {
			idx = got;
			if (qu(rq, RQ_PIX, idx) == kInactive) { T.cur = kEmpty; T.best = RT_NO_HIT; T.tmin = 0; T.bestTag = 0; }   // padding slot of an edge tile
			else travInit<false>(S, T, mk3(qf(rq, RQ_OX, idx), qf(rq, RQ_OY, idx), qf(rq, RQ_OZ, idx)), mk3(qf(rq, RQ_DX, idx), qf(rq, RQ_DY, idx), qf(rq, RQ_DZ, idx)));
		}
		if (__ballot(idx != kInactive) == 0) { if (workLeft(pool)) continue; break; }
		for (;;) {
			if (idx != kInactive) {
				if (T.cur != kEmpty) travRound<false, CSG, COUNT>(S, stack, T);
				if (T.cur == kEmpty) {   // bvh.cpp:91: return hitObject
					qsu(hq, HQ_PRIM, idx, T.best);
					qsf(hq, HQ_T, idx, T.best == RT_NO_HIT ? 0.0f : T.tmin);
					if (CSG) qsu(hq, HQ_TAG, idx, T.bestTag);
					if (COUNT) { cn += T.cntNodes; cp += T.cntPrims; atomicMax(&ctr->pad, (uint32_t)T.cntNodes); }
					idx = kInactive;
				}
			}
			const int busy = __popcll(__ballot(idx != kInactive));
			if (busy == 0 || (workLeft(pool) && busy < kRefillBelow)) break;
		}
	}
	if (COUNT) { atomicAdd(&ctr->nodes_nearest, cn); atomicAdd(&ctr->prims_nearest, cp); }
}

// Packet hand-out.  Every wave starts on the packet with its own number and asks the device-wide cursor for the
// ones after those, one atomic per 64 rays, issued as soon as the walk of the current packet is over.  (A cursor hit
// by 8 000 waves in the same microsecond serves ~90 of them per microsecond: with the first packet handed out by the
// cursor too, that stampede cost each wave ~45 us of a 650 us kernel, and because loads and returning atomics come
// back in order it cannot be hidden behind other memory work.)  Measured alternatives, both dropped: cutting the
// range into eight per-XCD parts so that each L2 sees one band of the image was 1.8x SLOWER on the C2 frame, and
// handing packets out in a scattered order to avoid neighbouring waves walking the same lines gained only 5 %.
// Packets are handed out kPacketChunk at a time: packet numbers wave*chunk .. +chunk-1 statically, then chunks from
// the cursor.  PacketFeed walks through the chunk a wave holds.
// Static share (round 2): with the quad walk a packet takes ~25 us and the 2 048 waves of a launch came to the cursor 66 times per
// microsecond, close to what one address serves (~90): every wave's request queued behind the others'.  Now RT_PACKET_STATIC_PCT of
// the packets are handed out without the cursor -- wave w takes packets w, w + W, w + 2 W, ... (W = waves of the launch; the hand-out
// order already mixes cheap and expensive packets) -- and only the rest, which evens out the waves' different luck, goes through it.
// Measured (share 0 / 50 / 75 / 90 %): nearest-hit kernel alone 0.497 / 0.458 / 0.469 / 0.487 ms, shadow kernel ~0.49 ms throughout; with
// four lanes in flight the frame rate is the same within noise (the other lanes' kernels were filling those waits already).
#ifndef RT_PACKET_STATIC_PCT
#define RT_PACKET_STATIC_PCT 50
#endif
struct PacketFeed { uint32_t j, S; };   // static packets taken so far / this wave's static share
__device__ __forceinline__ PacketFeed firstPackets(uint32_t nPackets) {
	const uint32_t W = waveCount();
	PacketFeed f;
	f.j = 0;
	uint32_t S = (uint32_t)((unsigned long long)nPackets * RT_PACKET_STATIC_PCT / 100u / W);
	if (S < 1u) S = 1u;
	f.S = S;
	return f;
}
__device__ __forceinline__ uint32_t fetchPacket(PacketFeed& f, uint32_t* head, uint32_t nPackets) {
	const uint32_t W = waveCount();
	if (f.j < f.S) {   // (wave-uniform)
		const unsigned long long b = (unsigned long long)waveNumber() + (unsigned long long)f.j * W;
		f.j++;
		if (b < nPackets) return (uint32_t)b;
		f.j = f.S;
	}
	if ((unsigned long long)nPackets <= (unsigned long long)W * f.S) return kInactive;   // every packet was handed out statically
	uint32_t k = 0;
	if (__lane_id() == 0) k = atomicAdd(head, 1u);
	const unsigned long long b = (unsigned long long)__builtin_amdgcn_readfirstlane(k) + (unsigned long long)W * f.S;
	return b < nPackets ? (uint32_t)b : kInactive;
}

template <bool CSG, bool COUNT>
__global__ __launch_bounds__(kBlock, RT_PACKET_WAVES_PER_SIMD) void k_intersect_packet(DevScene S, RtQueue rq, uint32_t nArg, const uint32_t* nDev, RtQueue hq, RtCounters* ctr, uint32_t groupStride) {
	const uint32_t lane = __lane_id();
	const uint32_t n = deviceCount(nArg, nDev);
	const PacketLds ws = packetLdsInit(S);
	// Hand-out order: groups of kPacketGroup neighbouring packets, the groups visited with a stride coprime to their
	// number.  A frame has stretches of packets that cost nothing (sky: every ray misses the root box); handed out in
	// image order they reach the cursor faster than it can serve them (~90 atomics/us) and the end of the launch ran at
	// the cursor's pace.  Strided, cheap and expensive packets are in flight together all the time.
	const uint32_t nGroups = ((n + 63u) / 64u + kPacketGroup - 1u) / kPacketGroup;
	const uint32_t nPackets = nGroups * kPacketGroup;   // the last group may hold empty packets
	// (groupStride 0: in order -- the launches of deeper levels, whose ray count only the device knows; no sky stretches there)
	auto permute = [&](uint32_t k) {
		if (k == kInactive || groupStride == 0u) return k;
		const uint32_t g = (uint32_t)(((unsigned long long)(k / kPacketGroup) * groupStride) % nGroups);
		return g * kPacketGroup + k % kPacketGroup;
	};
	unsigned long long cn = 0, cp = 0, cs = 0;
	TR_DECL(0)
	struct Rays { f3 e, d; bool alive; };
	auto loadRays = [&](uint32_t pk) {
		Rays r;
		r.e = r.d = mk3(0, 0, 0); r.alive = false;
		if (pk != kInactive) {
			const uint32_t idx = pk * 64u + lane;
			if (idx < n && qu(rq, RQ_PIX, idx) != kInactive) {
				r.alive = true;
				r.e = mk3(qf(rq, RQ_OX, idx), qf(rq, RQ_OY, idx), qf(rq, RQ_OZ, idx));
				r.d = mk3(qf(rq, RQ_DX, idx), qf(rq, RQ_DY, idx), qf(rq, RQ_DZ, idx));
			}
		}
		return r;
	};
	PacketFeed feed = firstPackets(nPackets);
	uint32_t pk = permute(fetchPacket(feed, &ctr->head_nearest, nPackets));
	while (pk != kInactive) {
		TR(1, pk);
		const Rays cur = loadRays(pk);
		const uint32_t idx = pk * 64u + lane;
		Trav T;
		T.cur = kEmpty; T.best = RT_NO_HIT; T.tmin = 0; T.bestTag = 0; T.cntNodes = 0; T.cntPrims = 0; T.cntSteps = 0;
		if (cur.alive) travInit<false>(S, T, cur.e, cur.d);
		TR(2, __popcll(__ballot(T.cur != kEmpty)));
		packetTraverse<false, CSG, COUNT>(S, ws, T, cur.alive TR_ARG);
		TR(9, 0);
		pk = permute(fetchPacket(feed, &ctr->head_nearest, nPackets));
		TR(10, 0);
		if (idx < n) {
			qsu(hq, HQ_PRIM, idx, cur.alive ? T.best : RT_NO_HIT);
			qsf(hq, HQ_T, idx, (cur.alive && T.best != RT_NO_HIT) ? T.tmin : 0.0f);
			if (CSG) qsu(hq, HQ_TAG, idx, T.bestTag);
		}
		if (COUNT) { cn += T.cntNodes; cp += T.cntPrims; cs += T.cntSteps; }
		TR(11, 0);
		WL_PACKET(T, cur.alive);
	}
	WL_END();
	if (COUNT) { atomicAdd(&ctr->nodes_nearest, cn); atomicAdd(&ctr->prims_nearest, cp); if (lane == 0) atomicAdd(&ctr->steps_nearest, cs); }
}

// hit queue -> caller-visible (hit, id, t)
__global__ __launch_bounds__(kBlock) void k_export_hits(DevScene S, RtQueue hq, uint32_t n, uint8_t* hit, int32_t* id, float* t) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const uint32_t prim = qu(hq, HQ_PRIM, i);
	const bool h = prim != RT_NO_HIT;
	if (hit) hit[i] = h;
	if (id) id[i] = h ? primId(S, prim) : -1;
	if (t) t[i] = h ? qf(hq, HQ_T, i) : 0.0f;
}

// rt_trace_pick: one event of the narration (include/rt_capi.h)
__device__ __forceinline__ void pickEmit(const RtPick& P, uint32_t path, int kind, int arg, int flags, f3 p) {
	const uint32_t i = atomicAdd(P.count, 1u);
	if (i < P.cap) {
		rt_pick_event e;
		e.path = path; e.kind = kind; e.arg = arg; e.flags = flags; e.p[0] = p.x; e.p[1] = p.y; e.p[2] = p.z;
		P.ev[i] = e;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Object::transmitRay objects.cpp:107-140 (+ Plane:: :224-229, Triangle:: :273-275).
// Where the exit search on the ray's own object fails, the reference continues from whatever an earlier ray left in
// the object's hit cache; like the oracle restatement this uses the current point / previous normal instead.
// ---------------------------------------------------------------------------------------------------------
template <bool CSG, bool PICK, class Store, bool TRIS = true>
__device__ inline bool transmitRay(const DevScene& S, uint32_t ref, float refraction, f3 inPoint, f3 inVector, f3 inNormal, f3& outPoint, f3& outVector, Store& csgStore,
                                   const RtPick& pick, uint32_t path, uint32_t& failedSearches) {
	const uint32_t type = RT_PRIM_TYPE(ref), ix = RT_PRIM_INDEX(ref);
	if (type == RT_PRIM_PLANE || refraction == 0) {
		outVector = inVector;
		outPoint = inPoint + 1e-5f * outVector;
		return true;
	}
	f3 currentPoint = inPoint, innerVector = mk3(0, 0, 0), outNormal = mk3(0, 0, 0);
	refractRay(inVector, inNormal, refraction, innerVector);
	rt_csg c = {};
	if (CSG && type == RT_PRIM_CSG) {
		c = S.csgs[ix]; csgStore.loadProgram(S, c);
		// csgObject::getNormal returns the member cachedHitNormal, which a failed exit search leaves alone: the normal of the last
		// successful rayhit on the object, i.e. of the hit this ray entered through (the other classes write into transmitRay's local)
		outNormal = inNormal;
	}
	bool result = false;
	int reflections = 0;
	while (!result && reflections < RT_MAX_REFLECTIONS) {
		f3 hp = currentPoint;
		float tExit = 0.0f;   // (what the oracle counts as a hidden-state case: an exit search with no exit, restate.cpp transmitRay)
		if (type == RT_PRIM_SPHERE) {
			const rt_sphere s = S.spheres[ix];
			const float t = sphereHit(ld3(s.center), s.radius, currentPoint, innerVector, true);
			tExit = t;
			// cachedHitpoint is rewritten whenever rayhit gets past its two early returns; t == 0 can be either
			// "missed" (no write) or "exactly at the origin" (write of e + 0*d = e): both leave hp == currentPoint.
			if (t > 0.0f) hp = currentPoint + t * innerVector;
			outNormal = sphereNormal(S, s, hp);
		} else if (TRIS && type == RT_PRIM_TRI) {   // (TRIS false: the caller knows that no triangle of the scene transmits)
			f3 nrm = outNormal;
			const float t = meshHit(S, S.tris[ix].mesh, currentPoint, innerVector, true, nrm);
			if (t > 0.0f) hp = currentPoint + t * innerVector;
			outNormal = nrm;
			tExit = t;
		} else if (type == RT_PRIM_BOX) {
			const rt_box b = S.boxes[ix];
			const float t = boxIntersect(b.b, currentPoint, innerVector, true);
			if (!(t < 0)) hp = currentPoint + t * innerVector;
			boxNormal(b.b, hp, outNormal);
			tExit = t;
		} else if (CSG) {
			uint32_t tag = 0;
			const float t = csgEval(S, c, currentPoint, innerVector, true, tag, csgStore);
			hp = currentPoint + t * innerVector;   // csgObject::rayhit always rewrites cachedHitpoint (csg.cpp:24)
			if (t != 0.0f) outNormal = csgNormal(S, c, currentPoint, innerVector, tag, csgStore);
			tExit = t;
		}
		// objects.cpp:123-125: the reference goes on from whatever the object's hit cache holds -- for a top-level mesh that is state left by
		// EARLIER pixels, which the device does not have.  Counted, so that a caller can report the fraction.
		if (!(tExit > 0.0f)) failedSearches++;
		outPoint = hp;
		result = refractRay(innerVector, outNormal, refraction, outVector);
		if (!result) {
			if (PICK) pickEmit(pick, path, RT_PICK_INTERIOR, reflections, 0, outPoint);   // objects.cpp:129-131
			innerVector = reflectRay(-innerVector, outNormal);
			currentPoint = outPoint;
			reflections++;
		}
	}
	return result;
}

// What Object::lightPoint (objects.cpp:54-105) does for one ray apart from tracing its children and its shadow rays: the background a
// miss fetches, normal, material (texel), ambient term, the reflected child and the request for the transmitted one.  Shared by
// k_shade and k_level; forced inline, the outputs are the callers' registers.
struct ShadeOut {
	f3 own, beyond; bool hasOwn, hasBeyond;             // what the ray adds to its pixel by itself (MULTI: sent by the caller)
	bool wantR, wantX, wantS, wantTex; uint32_t texTri; float refraction;
	f3 p, N, V, rO, rD, rW, tW, wl; float kd[3], ks[3], shininess;
};
template <bool CSG, bool PICK, bool TEXFIX, bool MULTI, class TagOf>
__device__ __forceinline__ void shadeRay(const DevScene& S, bool live, bool hit, uint32_t prim, float t, f3 e, f3 d, f3 w, uint32_t pixel, uint32_t path, f3 background,
                                         const RtAccum& A, int maxDepth, int wantSurfaces, const RtPick& pick, TagOf tagOf, ShadeOut& o) {
	f3& own = o.own; f3& beyond = o.beyond; bool& hasOwn = o.hasOwn; bool& hasBeyond = o.hasBeyond;
	bool& wantR = o.wantR; bool& wantX = o.wantX; bool& wantS = o.wantS; bool& wantTex = o.wantTex; uint32_t& texTri = o.texTri; float& refraction = o.refraction;
	f3& p = o.p; f3& N = o.N; f3& V = o.V; f3& rO = o.rO; f3& rD = o.rD; f3& rW = o.rW; f3& tW = o.tW; f3& wl = o.wl;
	float (&kd)[3] = o.kd; float (&ks)[3] = o.ks; float& shininess = o.shininess;
	// what this ray adds to its pixel by itself: the background it ran into (q1.cpp:143-145 / objects.cpp:68-69,94-95) or the ambient
	// term of its hit, and the background a reflected ray beyond the depth limit would have fetched; sent at the end of the kernel
	own = w * background; beyond = mk3(0, 0, 0);
	hasOwn = live && !hit; hasBeyond = false;
	if (!MULTI && hasOwn) accumAdd(A, pixel, own);
	if (PICK && live && !hit && path > 1u) pickEmit(pick, path, RT_PICK_MISS, 0, 0, mk3(0, 0, 0));

	wantR = false; wantX = false; wantS = false;
	wantTex = false;      // TEXFIX: a hit on a reflective textured triangle leaves a record (RtTexFix)
	texTri = 0;
	refraction = 0;
	p = mk3(0, 0, 0); N = mk3(0, 0, 0); V = mk3(0, 0, 0);
	rO = p; rD = p; rW = p; tW = p; wl = p;
	kd[0] = kd[1] = kd[2] = 0; ks[0] = ks[1] = ks[2] = 0; shininess = 0;
	if (hit) {
		const uint32_t type = RT_PRIM_TYPE(prim), ix = RT_PRIM_INDEX(prim);
		p = e + t * d;   // cachedHitpoint of the winner (every rayhit sets e + t*d)
		V = normalize(-d);
		const rt_material m = S.materials[primMaterial(S, prim)];
		float ka[3] = { m.ambient[0], m.ambient[1], m.ambient[2] };
		kd[0] = m.diffuse[0]; kd[1] = m.diffuse[1]; kd[2] = m.diffuse[2];
		ks[0] = m.specular[0]; ks[1] = m.specular[1]; ks[2] = m.specular[2];
		shininess = m.shininess;
		if (type == RT_PRIM_SPHERE) N = sphereNormal(S, S.spheres[ix], p);
		else if (type == RT_PRIM_PLANE) N = normalize(ld3(S.planes[ix].normal));   // objects.cpp:215-217
		else if (type == RT_PRIM_TRI) {
			const rt_tri tr = S.tris[ix];
			N = ld3(tr.n);
			if (tr.uv >= 0) {   // TextureTriangle::lightPoint, texturemesh.cpp:25-52
				const rt_uv uvs = S.uvs[tr.uv];
				const f3 P0 = ld3(tr.p0), P1 = ld3(tr.p1), P2 = ld3(tr.p2);
				const f3 v0 = P0 - p, v1 = P1 - p, v2 = P2 - p;
				const float area = length(cross(P0 - P1, P0 - P2));
				const float a0 = length(cross(v1, v2)) / area;
				const float a1 = length(cross(v2, v0)) / area;
				const float a2 = length(cross(v0, v1)) / area;
				const float u = (uvs.uv[0] * a0 + uvs.uv[2] * a1) + uvs.uv[4] * a2;
				const float v = (uvs.uv[1] * a0 + uvs.uv[3] * a1) + uvs.uv[5] * a2;
				const int tex = S.meshes[tr.mesh].tex;
				const int height = S.textures[tex].height, width = S.textures[tex].width;
				const uint8_t* px = texel(S, tex, f2i(u * width), f2i(v * height));   // texturemesh.cpp:7-16
				ka[0] = kd[0] = float(px[0]) / 255; ka[1] = kd[1] = float(px[1]) / 255; ka[2] = kd[2] = float(px[2]) / 255;
				if (TEXFIX && !isZero3(m.reflective)) { wantTex = true; texTri = ix; }
			}
		} else if (type == RT_PRIM_BOX) boxNormal(S.boxes[ix].b, p, N);
		else if (CSG) N = csgNormal(S, S.csgs[ix], e, d, tagOf());

		const bool transmissive = !isZero3(m.transmissive);
		const f3 kt = ld3(m.transmissive);
		if (PICK) pickEmit(pick, path, RT_PICK_HIT, (int)type, (isZero3(m.reflective) ? 0 : 1) | (transmissive ? 2 : 0), p);   // raytracer.cpp:323-324
		wl = transmissive ? w * (mk3(1.0f, 1.0f, 1.0f) - kt) : w;   // objects.cpp:103
		const int depth = 31 - __clz((int)path);   // reflectionCount of this ray

		// ambient lights, objects.cpp:382-387
		f3 amb = mk3(0, 0, 0);
		for (int l = 0; l < S.n_lights; l++)
			if (S.lights[l].type == RT_LIGHT_AMBIENT) amb = amb + ld3(S.lights[l].colour) * ld3(ka);
		if (MULTI) { own = wl * amb; hasOwn = true; } else accumAdd(A, pixel, wl * amb);

		if (!isZero3(m.reflective)) {   // objects.cpp:62-75
			const f3 R = reflectRay(V, N);
			rO = p + 1e-5f * R;
			rD = (p + R) - rO;
			rW = wl * ld3(m.reflective);
			if (depth + 1 > maxDepth) {   // raytracer.cpp:308-313 -> objects.cpp:68-69
				if (MULTI) { beyond = rW * background; hasBeyond = true; } else accumAdd(A, pixel, rW * background);
				if (PICK) pickEmit(pick, path * 2u, RT_PICK_MAXDEPTH, 0, 0, mk3(0, 0, 0));
			} else wantR = true;
		}
		// objects.cpp:81-104: the search for the exit point (Object::transmitRay: up to 16 internal reflections, each a full
		// hit test of the object) is not done here: a few such rays would hold their whole wave.  They are queued, and
		// k_transmit works through them with every lane busy.
		if (transmissive) { wantX = true; tW = w * kt; refraction = m.refraction; }
		wantS = wantSurfaces != 0;
	}

}

// ---------------------------------------------------------------------------------------------------------
// shade: everything Object::lightPoint does except the shadow rays
// ---------------------------------------------------------------------------------------------------------
// (the plain variant fits 64 VGPRs without spilling: 8 waves per SIMD, so that it shares a SIMD with whatever the other lanes run: +1 % on C2)
// MULTI: several samples per pixel -- the accumulator atomics of neighbouring lanes are combined first (accumAddRows)
template <bool CSG, bool PICK, bool TEXFIX, bool MULTI>
__global__ __launch_bounds__(kBlock, (CSG || PICK || TEXFIX || MULTI) ? 1 : 8) void k_shade(DevScene S, RtQueue rq, RtQueue hq, uint32_t nArg, const uint32_t* nDev, RtQueue nq, RtQueue sq, RtQueue tq, RtCounters* ctr,
                                                  RtAccum A, uint32_t aaSq, int maxDepth, int wantSurfaces,
                                                  uint8_t* outHit, int32_t* outId, float* outT, RtPick pick, RtTexFix tex) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t n = deviceCount(nArg, nDev);
	if (blockIdx.x * kBlock >= n) return;   // (workgroup-uniform: a launch sized by a bound leaves its surplus workgroups at once)
	// every record is read at once (index clamped, result masked): one memory round trip instead of two
	const uint32_t ic = i < n ? i : 0u;
	uint32_t pixKey = qu(rq, RQ_PIX, ic), path = qu(rq, RQ_PATH, ic), prim = qu(hq, HQ_PRIM, ic);
	f3 e = mk3(qf(rq, RQ_OX, ic), qf(rq, RQ_OY, ic), qf(rq, RQ_OZ, ic));
	f3 d = mk3(qf(rq, RQ_DX, ic), qf(rq, RQ_DY, ic), qf(rq, RQ_DZ, ic));
	f3 w = mk3(qf(rq, RQ_WR, ic), qf(rq, RQ_WG, ic), qf(rq, RQ_WB, ic));
	float t = qf(hq, HQ_T, ic);
	const bool live = i < n && pixKey != kInactive;
	if (!live) { pixKey = 0; path = 1; prim = RT_NO_HIT; e = mk3(0, 0, 0); d = mk3(0, 0, 1); w = mk3(0, 0, 0); t = 0; }
	const uint32_t pixel = pixKey / aaSq;
	const bool hit = live && prim != RT_NO_HIT;
	const f3 background = ld3(S.background);
	if (live && path == 1u && outHit) {   // first-level results for rt_trace_batch
		outHit[pixKey] = hit;
		if (outId) outId[pixKey] = hit ? primId(S, prim) : -1;
		if (outT) outT[pixKey] = hit ? t : 0.0f;
	}
	ShadeOut o;
	shadeRay<CSG, PICK, TEXFIX, MULTI>(S, live, hit, prim, t, e, d, w, pixel, path, background, A, maxDepth, wantSurfaces, pick, [&] { return qu(hq, HQ_TAG, i); }, o);
	const bool wantR = o.wantR, wantX = o.wantX, wantS = o.wantS, wantTex = o.wantTex, hasOwn = o.hasOwn, hasBeyond = o.hasBeyond;
	const uint32_t texTri = o.texTri; const float refraction = o.refraction, shininess = o.shininess;
	const f3 own = o.own, beyond = o.beyond, p = o.p, N = o.N, V = o.V, rO = o.rO, rD = o.rD, rW = o.rW, tW = o.tW, wl = o.wl;
	const float* kd = o.kd; const float* ks = o.ks;

	// ---- convergent part: compacted appends, ONE atomic per counter per workgroup.  An append counter is a single
	// address, and a single address takes ~90 atomics per microsecond on this GPU: with one atomic per wave (32 400 waves
	// a frame) that rate, not memory or arithmetic, set the time of this kernel. ----
	uint32_t sr, st, ss;
	{
		__shared__ uint32_t waveCount[kBlock / 64][3];
		__shared__ uint32_t blockBase[3];
		const unsigned long long mR = __ballot(wantR), mT = __ballot(wantX), mS = __ballot(wantS);
		const uint32_t lane = __lane_id(), wv = threadIdx.x >> 6;
		if (lane == 0) { waveCount[wv][0] = (uint32_t)__popcll(mR); waveCount[wv][1] = (uint32_t)__popcll(mT); waveCount[wv][2] = (uint32_t)__popcll(mS); }
		__syncthreads();
		uint32_t beforeN = 0, beforeS = 0, totalN = 0, totalS = 0;   // rays / surfaces of the waves before this one, and of the workgroup
		uint32_t beforeX = 0, totalX = 0;
		for (uint32_t k = 0; k < kBlock / 64; k++) {
			const uint32_t cn = waveCount[k][0], cx = waveCount[k][1], cs = waveCount[k][2];
			if (k < wv) { beforeN += cn; beforeX += cx; beforeS += cs; }
			totalN += cn; totalX += cx; totalS += cs;
		}
		if (threadIdx.x == 0 && totalN > 0) blockBase[0] = atomicAdd(&ctr->next_count, totalN);
		if (threadIdx.x == 64 && totalS > 0) blockBase[1] = atomicAdd(&ctr->surf_count, totalS);
		if (threadIdx.x == 128 && totalX > 0) blockBase[2] = atomicAdd(&ctr->trans_count, totalX);
		__syncthreads();
		const unsigned long long below = (1ull << lane) - 1ull;
		sr = blockBase[0] + beforeN + (uint32_t)__popcll(mR & below);
		st = blockBase[2] + beforeX + (uint32_t)__popcll(mT & below);
		ss = blockBase[1] + beforeS + (uint32_t)__popcll(mS & below);
	}
	if (wantR) {
		if (sr < nq.cap) {
			qsf(nq, RQ_OX, sr, rO.x); qsf(nq, RQ_OY, sr, rO.y); qsf(nq, RQ_OZ, sr, rO.z);
			qsf(nq, RQ_DX, sr, rD.x); qsf(nq, RQ_DY, sr, rD.y); qsf(nq, RQ_DZ, sr, rD.z);
			qsf(nq, RQ_WR, sr, rW.x); qsf(nq, RQ_WG, sr, rW.y); qsf(nq, RQ_WB, sr, rW.z);
			qsu(nq, RQ_PIX, sr, pixKey); qsu(nq, RQ_PATH, sr, path * 2u);
		} else ctr->overflow = 1;
	}
	if (wantX) {
		if (st < tq.cap) {
			qsu(tq, TQ_PRIM, st, prim);
			qsf(tq, TQ_PX, st, p.x); qsf(tq, TQ_PY, st, p.y); qsf(tq, TQ_PZ, st, p.z);
			qsf(tq, TQ_DX, st, d.x); qsf(tq, TQ_DY, st, d.y); qsf(tq, TQ_DZ, st, d.z);
			qsf(tq, TQ_NX, st, N.x); qsf(tq, TQ_NY, st, N.y); qsf(tq, TQ_NZ, st, N.z);
			qsf(tq, TQ_REFR, st, refraction);
			qsf(tq, TQ_WR, st, tW.x); qsf(tq, TQ_WG, st, tW.y); qsf(tq, TQ_WB, st, tW.z);
			qsu(tq, TQ_PIX, st, pixKey); qsu(tq, TQ_PATH, st, path);
		} else ctr->overflow = 1;
	}
	if (TEXFIX) {
		const uint32_t slot = wavePush(wantTex, tex.count);
		if (wantTex) {
			if (slot < tex.cap) {
				RtTexRec r;
				r.pixKey = pixKey; r.path = path; r.tri = texTri;
				r.c[0] = kd[0]; r.c[1] = kd[1]; r.c[2] = kd[2];
				r.wl[0] = wl.x; r.wl[1] = wl.y; r.wl[2] = wl.z;
				r.p[0] = p.x; r.p[1] = p.y; r.p[2] = p.z;
				r.N[0] = N.x; r.N[1] = N.y; r.N[2] = N.z;
				r.next = atomicExch(&tex.heads[pixel], slot);
				tex.rec[slot] = r;
			} else ctr->overflow = 1;
		}
	}
	if (wantS) {
		if (ss < sq.cap) {
			qsf(sq, SQ_PX, ss, p.x); qsf(sq, SQ_PY, ss, p.y); qsf(sq, SQ_PZ, ss, p.z);
			qsf(sq, SQ_NX, ss, N.x); qsf(sq, SQ_NY, ss, N.y); qsf(sq, SQ_NZ, ss, N.z);
			qsf(sq, SQ_VX, ss, V.x); qsf(sq, SQ_VY, ss, V.y); qsf(sq, SQ_VZ, ss, V.z);
			qsf(sq, SQ_KDR, ss, kd[0]); qsf(sq, SQ_KDG, ss, kd[1]); qsf(sq, SQ_KDB, ss, kd[2]);
			qsf(sq, SQ_KSR, ss, ks[0]); qsf(sq, SQ_KSG, ss, ks[1]); qsf(sq, SQ_KSB, ss, ks[2]);
			qsf(sq, SQ_SHIN, ss, shininess);
			qsf(sq, SQ_WR, ss, wl.x); qsf(sq, SQ_WG, ss, wl.y); qsf(sq, SQ_WB, ss, wl.z);
			qsu(sq, SQ_PIX, ss, pixKey); qsu(sq, SQ_PATH, ss, path);
		} else ctr->overflow = 1;
	}
	if (MULTI) accumAddRows(A, pixel, own, hasOwn, beyond, hasBeyond);
}

// k_level: an append that found its queue full.  A returning atomic whose result is waited for, so that the flag is set at the memory
// side before the lane goes on (the level's report, written from the same launch, must not miss it).
__device__ __forceinline__ void levelOverflow(RtCounters* ctr) { const uint32_t was = atomicExch(&ctr->overflow, 1u); asm volatile("" :: "v"(was)); }

// ---------------------------------------------------------------------------------------------------------
// One kernel per DEEPER level (round 3).  Below level 0 a level of the bundled scenes is a few hundred thousand incoherent rays at
// most, and walking, shading and lighting them in four launches (k_intersect, k_shade, k_publish, k_shadow_point) cost 50 us per
// level of which most was launch-to-launch latency and the queues in between (hit records, 84 B surface records): scene a, seventeen
// levels of mirrors, spent 0.8 of its 1.0 ms there.  Here one lane takes one ray through all of it: BVH::findNearest (the per-lane
// walk), Object::lightPoint's own part (shadeRay), the shadow rays towards the point-type lights (per-lane any-hit walks, as in
// k_shadow_point) -- hit and surface never leave the registers -- and the compacted appends of its children.  Exit searches of
// transmitted rays stay in k_transmit (a few trapped rays would hold their waves), area lights in k_shadow_area (their samples fill
// a wave per surface): for those the requests / surface records are written as k_shade writes them.  Same device functions, same
// arithmetic, integer pixel sums: frames are bit-identical to the four-kernel form (Run::levels keeps that form for counting and
// timing runs, and the tests compare the two).
// `surfaces`: 0 = the scene has no lights that cast shadow rays, 1 = count the hits (rt_stats), 2 = also write their records.
__device__ __forceinline__ void pointLightGeometry(const rt_light& L, f3 p, f3& Ld, f3& lightPos);
//
// SMALL: the scene's arrays that the walks and the shading read -- tree, leaf lists, primitives, planes, materials, lights, CSG programs:
// RtSmallScene, one blob assembled at upload when they come to a few tens of KB, which holds for every scene the reference ships -- are
// copied into the workgroup's LDS first and the DevScene the device functions see points there.  A ray of such a level is a chain of
// some thirty dependent loads (node, leaf entry, primitive, ... for the nearest hit and again for every light) at 0.5 - 1 us each from
// the caches: that chain, not arithmetic or bandwidth, was the length of the launch.  From LDS a link costs a tenth.  The per-lane
// stack comes out of the same dynamic LDS, sized by the tree's depth.
//
// XINLINE: the exit search of a transmitted ray (Object::transmitRay) happens here as well, right after the shading, instead of in a
// k_transmit launch behind this one.  Only for scenes whose transmitting objects are spheres, boxes and planes -- one analytic exit
// test per internal reflection; a CSG object or a mesh would hold the wave for a walk or an evaluation per reflection (rt_scene_upload
// decides).  `report` (may be null): the level's counts for the host loop, written by the LAST workgroup to finish its appends (a
// ticket per workgroup with work: a few hundred at most at these sizes) -- with that a deeper level is ONE launch instead of the
// k_level / k_transmit / k_publish chain; a level with a k_transmit behind it leaves the report to k_publish.
template <bool CSG, bool SMALL, bool XINLINE>
__global__ __launch_bounds__(kBlock) void k_level(DevScene Sg, RtSmallScene small, RtQueue rq, uint32_t nArg, const uint32_t* nDev, RtQueue nq, RtQueue sq, RtQueue tq, RtCounters* ctr,
                                                  RtAccum A, uint32_t aaSq, int maxDepth, int surfaces, const int32_t* lightList, uint32_t nLights,
                                                  uint32_t* stateCases, RtLevelReport* report, uint32_t seq) {
	// (CSG && XINLINE: every kind of object, CSG evaluations and mesh walks included -- chosen where the levels are latency-bound, see rtk_level)
	extern __shared__ uint4 levelLds[];
	__shared__ int32_t stackStatic[SMALL ? 1 : kStack][kBlock];
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t n = deviceCount(nArg, nDev);
	if (blockIdx.x * kBlock >= n) return;   // (workgroup-uniform: a launch sized by a bound leaves its surplus workgroups at once)
	DevScene S = Sg;
	int32_t (*stack)[kBlock] = stackStatic;
	if (SMALL) {
		stack = reinterpret_cast<int32_t (*)[kBlock]>(levelLds);
		uint4* blob = levelLds + small.stackRows * (kBlock * 4u / 16u);
		for (uint32_t k = threadIdx.x; k < small.bytes / 16u; k += kBlock) blob[k] = small.blob[k];
		__syncthreads();
		const char* base = reinterpret_cast<const char*>(blob);
		S.treelets = reinterpret_cast<const float4*>(base + small.off[RT_SMALL_TREELETS]);
		S.leaf_prims = reinterpret_cast<const uint32_t*>(base + small.off[RT_SMALL_LEAF_PRIMS]);
		S.spheres = reinterpret_cast<const rt_sphere*>(base + small.off[RT_SMALL_SPHERES]);
		S.boxes = reinterpret_cast<const rt_box*>(base + small.off[RT_SMALL_BOXES]);
		S.planes = reinterpret_cast<const rt_plane*>(base + small.off[RT_SMALL_PLANES]);
		S.materials = reinterpret_cast<const rt_material*>(base + small.off[RT_SMALL_MATERIALS]);
		S.lights = reinterpret_cast<const rt_light*>(base + small.off[RT_SMALL_LIGHTS]);
		S.csgs = reinterpret_cast<const rt_csg*>(base + small.off[RT_SMALL_CSGS]);
		S.csg_nodes = reinterpret_cast<const rt_csg_node*>(base + small.off[RT_SMALL_CSG_NODES]);
		if (small.off[RT_SMALL_TRIS] != 0xffffffffu) S.tris = reinterpret_cast<const rt_tri*>(base + small.off[RT_SMALL_TRIS]);
		lightList = reinterpret_cast<const int32_t*>(base + small.off[RT_SMALL_LIGHT_LIST]);
	}
	const uint32_t ic = i < n ? i : 0u;
	uint32_t pixKey = qu(rq, RQ_PIX, ic), path = qu(rq, RQ_PATH, ic);
	f3 e = mk3(qf(rq, RQ_OX, ic), qf(rq, RQ_OY, ic), qf(rq, RQ_OZ, ic));
	f3 d = mk3(qf(rq, RQ_DX, ic), qf(rq, RQ_DY, ic), qf(rq, RQ_DZ, ic));
	f3 w = mk3(qf(rq, RQ_WR, ic), qf(rq, RQ_WG, ic), qf(rq, RQ_WB, ic));
	const bool live = i < n && pixKey != kInactive;
	if (!live) { pixKey = 0; path = 1; e = mk3(0, 0, 0); d = mk3(0, 0, 1); w = mk3(0, 0, 0); }
	// BVH::findNearest, bvh.cpp:72-121
	Trav T;
	T.cur = kEmpty; T.sp = 0; T.best = RT_NO_HIT; T.tmin = 0; T.bestTag = 0; T.cntNodes = 0; T.cntPrims = 0; T.cntSteps = 0;
	if (live) travInit<false>(S, T, e, d);
	while (T.cur != kEmpty) travRound<false, CSG, false>(S, stack, T);
	const uint32_t prim = live ? T.best : RT_NO_HIT;
	const bool hit = prim != RT_NO_HIT;
	const float t = hit ? T.tmin : 0.0f;
	const uint32_t bestTag = T.bestTag;
	const uint32_t pixel = pixKey / aaSq;
	// Object::lightPoint, objects.cpp:54-105
	ShadeOut o;
	shadeRay<CSG, false, false, false>(S, live, hit, prim, t, e, d, w, pixel, path, ld3(S.background), A, maxDepth, surfaces, RtPick{ nullptr, nullptr, 0 }, [&] { return bestTag; }, o);
	// Object::transmitRay, objects.cpp:81-104 (k_transmit's body)
	bool wantT = o.wantX;   // without XINLINE: a request for k_transmit; with it: a transmitted child for the next level
	f3 tO = mk3(0, 0, 0), tD = tO;
	if (XINLINE && o.wantX) {
		f3 outPoint, outVector;
		uint32_t failedSearches = 0;
		CsgScratch store;
		const bool ok = transmitRay<CSG, false, CsgScratch>(S, prim, o.refraction, o.p, d, o.N, outPoint, outVector, store, RtPick{ nullptr, nullptr, 0 }, path, failedSearches);
		if (failedSearches != 0u) { const uint32_t was = atomicAdd(stateCases, failedSearches); asm volatile("" :: "v"(was)); }   // (returned: complete before the workgroup's ticket)
		wantT = false;
		if (ok) {
			tO = outPoint;
			tD = (outPoint + outVector) - outPoint;
			const int depth = 31 - __clz((int)path);
			if (depth + 1 > maxDepth) accumAdd(A, pixel, o.tW * ld3(S.background));   // raytracer.cpp:308-313 -> objects.cpp:94-95
			else wantT = true;
		}
	}
	// compacted appends, one atomic per counter per workgroup (see k_shade)
	uint32_t sr, st, ss;
	{
		__shared__ uint32_t waveCount[kBlock / 64][3];
		__shared__ uint32_t blockBase[3];
		const unsigned long long mR = __ballot(o.wantR), mT = __ballot(wantT), mS = __ballot(o.wantS);
		const uint32_t lane = __lane_id(), wv = threadIdx.x >> 6;
		if (lane == 0) { waveCount[wv][0] = (uint32_t)__popcll(mR); waveCount[wv][1] = (uint32_t)__popcll(mT); waveCount[wv][2] = (uint32_t)__popcll(mS); }
		__syncthreads();
		uint32_t beforeN = 0, beforeS = 0, totalN = 0, totalS = 0, beforeX = 0, totalX = 0;
		for (uint32_t k = 0; k < kBlock / 64; k++) {
			const uint32_t cn = waveCount[k][0], cx = waveCount[k][1], cs = waveCount[k][2];
			if (k < wv) { beforeN += cn; beforeX += cx; beforeS += cs; }
			totalN += cn; totalX += cx; totalS += cs;
		}
		// (XINLINE: the transmitted children go into the next level's queue behind the workgroup's reflected ones)
		if (threadIdx.x == 0 && totalN + (XINLINE ? totalX : 0u) > 0) blockBase[0] = atomicAdd(&ctr->next_count, totalN + (XINLINE ? totalX : 0u));
		if (threadIdx.x == 64 && totalS > 0) blockBase[1] = atomicAdd(&ctr->surf_count, totalS);
		if (!XINLINE && threadIdx.x == 128 && totalX > 0) blockBase[2] = atomicAdd(&ctr->trans_count, totalX);
		__syncthreads();
		const unsigned long long below = (1ull << lane) - 1ull;
		sr = blockBase[0] + beforeN + (uint32_t)__popcll(mR & below);
		st = (XINLINE ? blockBase[0] + totalN : blockBase[2]) + beforeX + (uint32_t)__popcll(mT & below);
		ss = blockBase[1] + beforeS + (uint32_t)__popcll(mS & below);
	}
	if (XINLINE && wantT) {
		if (st < nq.cap) {
			qsf(nq, RQ_OX, st, tO.x); qsf(nq, RQ_OY, st, tO.y); qsf(nq, RQ_OZ, st, tO.z);
			qsf(nq, RQ_DX, st, tD.x); qsf(nq, RQ_DY, st, tD.y); qsf(nq, RQ_DZ, st, tD.z);
			qsf(nq, RQ_WR, st, o.tW.x); qsf(nq, RQ_WG, st, o.tW.y); qsf(nq, RQ_WB, st, o.tW.z);
			qsu(nq, RQ_PIX, st, pixKey); qsu(nq, RQ_PATH, st, path * 2u + 1u);
		} else levelOverflow(ctr);
	}
	if (o.wantR) {
		if (sr < nq.cap) {
			qsf(nq, RQ_OX, sr, o.rO.x); qsf(nq, RQ_OY, sr, o.rO.y); qsf(nq, RQ_OZ, sr, o.rO.z);
			qsf(nq, RQ_DX, sr, o.rD.x); qsf(nq, RQ_DY, sr, o.rD.y); qsf(nq, RQ_DZ, sr, o.rD.z);
			qsf(nq, RQ_WR, sr, o.rW.x); qsf(nq, RQ_WG, sr, o.rW.y); qsf(nq, RQ_WB, sr, o.rW.z);
			qsu(nq, RQ_PIX, sr, pixKey); qsu(nq, RQ_PATH, sr, path * 2u);
		} else levelOverflow(ctr);
	}
	if (!XINLINE && wantT) {
		if (st < tq.cap) {
			qsu(tq, TQ_PRIM, st, prim);
			qsf(tq, TQ_PX, st, o.p.x); qsf(tq, TQ_PY, st, o.p.y); qsf(tq, TQ_PZ, st, o.p.z);
			qsf(tq, TQ_DX, st, d.x); qsf(tq, TQ_DY, st, d.y); qsf(tq, TQ_DZ, st, d.z);
			qsf(tq, TQ_NX, st, o.N.x); qsf(tq, TQ_NY, st, o.N.y); qsf(tq, TQ_NZ, st, o.N.z);
			qsf(tq, TQ_REFR, st, o.refraction);
			qsf(tq, TQ_WR, st, o.tW.x); qsf(tq, TQ_WG, st, o.tW.y); qsf(tq, TQ_WB, st, o.tW.z);
			qsu(tq, TQ_PIX, st, pixKey); qsu(tq, TQ_PATH, st, path);
		} else levelOverflow(ctr);
	}
	if (o.wantS && surfaces == 2) {   // area lights: k_shadow_area reads the surface from its record
		if (ss < sq.cap) {
			qsf(sq, SQ_PX, ss, o.p.x); qsf(sq, SQ_PY, ss, o.p.y); qsf(sq, SQ_PZ, ss, o.p.z);
			qsf(sq, SQ_NX, ss, o.N.x); qsf(sq, SQ_NY, ss, o.N.y); qsf(sq, SQ_NZ, ss, o.N.z);
			qsf(sq, SQ_VX, ss, o.V.x); qsf(sq, SQ_VY, ss, o.V.y); qsf(sq, SQ_VZ, ss, o.V.z);
			qsf(sq, SQ_KDR, ss, o.kd[0]); qsf(sq, SQ_KDG, ss, o.kd[1]); qsf(sq, SQ_KDB, ss, o.kd[2]);
			qsf(sq, SQ_KSR, ss, o.ks[0]); qsf(sq, SQ_KSG, ss, o.ks[1]); qsf(sq, SQ_KSB, ss, o.ks[2]);
			qsf(sq, SQ_SHIN, ss, o.shininess);
			qsf(sq, SQ_WR, ss, o.wl.x); qsf(sq, SQ_WG, ss, o.wl.y); qsf(sq, SQ_WB, ss, o.wl.z);
			qsu(sq, SQ_PIX, ss, pixKey); qsu(sq, SQ_PATH, ss, path);
		} else levelOverflow(ctr);
	}
	// The level's counts for the host (k_publish's job): every workgroup with work takes a ticket once its appends -- counters and a
	// possible overflow flag -- are done, the last one reads the totals and writes the report.  Before the shadow rays: the host, which
	// queues the next levels from these counts, hears of them that much earlier.
	if (report) {
		// No fences: an agent-scope release would write this XCD's L2 back (the eight L2s are not coherent with each other) once per
		// workgroup.  Everything the report reads is the target of device-scope atomics, which execute at the memory side, and every one
		// of them has RETURNED to its lane -- the three append counters before the barrier above, the overflow flag and the failed-search
		// count where they are set -- before the barrier below, hence before this workgroup's ticket; the last ticket reads them back
		// with atomics as well.  (The queues' contents need no such care: the next level is another launch.)
		__syncthreads();
		if (threadIdx.x == 0) {
			const uint32_t expected = (n + kBlock - 1u) / kBlock;
			if (atomicAdd(&ctr->reserved[0], 1u) == expected - 1u) {
				report->texrec_count = 0u;   // (scenes that keep texel records do not come here)
				report->state_cases = stateCases ? atomicAdd(stateCases, 0u) : 0u;
				report->next_count = atomicAdd(&ctr->next_count, 0u);
				report->surf_count = atomicAdd(&ctr->surf_count, 0u);
				report->overflow = atomicAdd(&ctr->overflow, 0u);
				__threadfence_system();
				__hip_atomic_store(&report->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
	// Directional / Point / Spot::lightPoint, objects.cpp:399-458 -> BVH::calcShadow, bvh.cpp:123-160: as k_shadow_point, on the surface in registers
	if (hit) {
		for (uint32_t l = 0; l < nLights; l++) {
			const rt_light L = S.lights[lightList[l]];
			f3 Ld, lightPos;
			pointLightGeometry(L, o.p, Ld, lightPos);
			Trav Ts;
			Ts.cur = kEmpty; Ts.sp = 0;
			travInit<true>(S, Ts, o.p, lightPos - o.p);   // bvh.cpp:124
			while (Ts.cur != kEmpty) travRound<true, CSG, false>(S, stack, Ts);
			if (Ts.lit && !(L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff))) {   // objects.cpp:451
				const f3 c = phong(ld3(L.colour) * Ts.shadow, o.kd, o.ks, o.shininess, o.N, Ld, o.V);
				accumAdd(A, pixel, o.wl * c);
			}
		}
	}
}

// Transmitted rays (objects.cpp:81-104): one request per lane.  The launch covers the most requests the level could have
// made; the count is read from the counters the shading kernel just wrote.  A request is a chain of up to 16 dependent
// exit searches and the slowest one sets the length of the launch, so what counts here is the latency of one search:
// workgroups are single waves and, for CSG objects, the interval lists live in LDS (LDSCSG) whenever the scene's
// largest program fits (CsgLds; rtk_transmit decides).
constexpr int kTransmitBlock = 64;
template <bool CSG, bool LDSCSG, bool PICK>
__global__ __launch_bounds__(kTransmitBlock) void k_transmit(DevScene S, RtQueue tq, RtQueue nq, RtCounters* ctr, uint32_t* stateCases, RtAccum A, uint32_t aaSq, int maxDepth, RtPick pick) {
	extern __shared__ __align__(16) uint8_t transmitLds[];
	const uint32_t i = blockIdx.x * kTransmitBlock + threadIdx.x;
	const uint32_t n = ctr->trans_count < tq.cap ? ctr->trans_count : tq.cap;
	if (blockIdx.x * kTransmitBlock >= n) return;   // wave-uniform: whole workgroups beyond the requests leave at once
	bool want = false;
	f3 tO = mk3(0, 0, 0), tD = tO, tW = tO;
	uint32_t pixKey = 0, path = 1;
	if (i < n) {
		const uint32_t prim = qu(tq, TQ_PRIM, i);
		const f3 p = mk3(qf(tq, TQ_PX, i), qf(tq, TQ_PY, i), qf(tq, TQ_PZ, i));
		const f3 d = mk3(qf(tq, TQ_DX, i), qf(tq, TQ_DY, i), qf(tq, TQ_DZ, i));
		const f3 N = mk3(qf(tq, TQ_NX, i), qf(tq, TQ_NY, i), qf(tq, TQ_NZ, i));
		tW = mk3(qf(tq, TQ_WR, i), qf(tq, TQ_WG, i), qf(tq, TQ_WB, i));
		pixKey = qu(tq, TQ_PIX, i); path = qu(tq, TQ_PATH, i);
		f3 outPoint, outVector;
		bool ok;
		uint32_t failedSearches = 0;
		if constexpr (CSG && LDSCSG) {
			CsgLds store = csgLdsStore(transmitLds, kTransmitBlock, (int)threadIdx.x, S.csg_cap);
			ok = transmitRay<CSG, PICK>(S, prim, qf(tq, TQ_REFR, i), p, d, N, outPoint, outVector, store, pick, path, failedSearches);
		} else {
			CsgScratch store;
			ok = transmitRay<CSG, PICK>(S, prim, qf(tq, TQ_REFR, i), p, d, N, outPoint, outVector, store, pick, path, failedSearches);
		}
		if (failedSearches != 0u) atomicAdd(stateCases, failedSearches);   // (rare: a handful per frame)
		if (PICK) pickEmit(pick, path, ok ? RT_PICK_EXIT : RT_PICK_LOST, 0, 0, ok ? outPoint : mk3(0, 0, 0));   // objects.cpp:90-92, 100-101
		if (ok) {
			tO = outPoint;
			tD = (outPoint + outVector) - outPoint;
			const int depth = 31 - __clz((int)path);
			if (depth + 1 > maxDepth) {   // raytracer.cpp:308-313 -> objects.cpp:94-95
				accumAdd(A, pixKey / aaSq, tW * ld3(S.background));
				if (PICK) pickEmit(pick, path * 2u + 1u, RT_PICK_MAXDEPTH, 0, 0, mk3(0, 0, 0));
			} else want = true;
		}
	}
	const uint32_t st = wavePush(want, &ctr->next_count);
	if (want) {
		if (st < nq.cap) {
			qsf(nq, RQ_OX, st, tO.x); qsf(nq, RQ_OY, st, tO.y); qsf(nq, RQ_OZ, st, tO.z);
			qsf(nq, RQ_DX, st, tD.x); qsf(nq, RQ_DY, st, tD.y); qsf(nq, RQ_DZ, st, tD.z);
			qsf(nq, RQ_WR, st, tW.x); qsf(nq, RQ_WG, st, tW.y); qsf(nq, RQ_WB, st, tW.z);
			qsu(nq, RQ_PIX, st, pixKey); qsu(nq, RQ_PATH, st, path * 2u + 1u);
		} else ctr->overflow = 1;
	}
}

struct Surface { f3 p, N, V, w; float kd[3], ks[3], shininess; uint32_t pixKey, path; };
__device__ __forceinline__ Surface loadSurface(const RtQueue& sq, uint32_t s) {
	Surface r;
	r.p = mk3(qf(sq, SQ_PX, s), qf(sq, SQ_PY, s), qf(sq, SQ_PZ, s));
	r.N = mk3(qf(sq, SQ_NX, s), qf(sq, SQ_NY, s), qf(sq, SQ_NZ, s));
	r.V = mk3(qf(sq, SQ_VX, s), qf(sq, SQ_VY, s), qf(sq, SQ_VZ, s));
	r.w = mk3(qf(sq, SQ_WR, s), qf(sq, SQ_WG, s), qf(sq, SQ_WB, s));
	r.kd[0] = qf(sq, SQ_KDR, s); r.kd[1] = qf(sq, SQ_KDG, s); r.kd[2] = qf(sq, SQ_KDB, s);
	r.ks[0] = qf(sq, SQ_KSR, s); r.ks[1] = qf(sq, SQ_KSG, s); r.ks[2] = qf(sq, SQ_KSB, s);
	r.shininess = qf(sq, SQ_SHIN, s);
	r.pixKey = qu(sq, SQ_PIX, s); r.path = qu(sq, SQ_PATH, s);
	return r;
}

// Directional / Point / Spot lights: one shadow ray per (surface, light); consecutive work items are consecutive
// surfaces of the SAME light so a wave's rays share their target.  Persistent with lane refill like k_intersect.
__device__ __forceinline__ void pointLightGeometry(const rt_light& L, f3 p, f3& Ld, f3& lightPos) {
	if (L.type == RT_LIGHT_DIRECTIONAL) {   // objects.cpp:399-410
		Ld = -ld3(L.direction);
		lightPos = p + RT_MAX_T * Ld;
	} else {
		lightPos = ld3(L.position);   // objects.cpp:422-431, 445-458
		Ld = normalize(lightPos - p);
	}
}

template <bool CSG, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_shadow_point(DevScene S, RtQueue sq, uint32_t nsurfArg, const int32_t* lightList, uint32_t nLights,
                                                         RtAccum A, uint32_t aaSq, RtCounters* ctr, int countOnDevice) {
	__shared__ int32_t stack[kStack][kBlock];
	const uint32_t nsurf = countOnDevice ? deviceCount(nsurfArg, &ctr->surf_count) : nsurfArg;
	if (nsurf == 0) return;
	const unsigned long long total = (unsigned long long)nsurf * nLights;
	uint32_t idx = kInactive;
	WorkPool pool = firstPool(total);
	Trav T;
	T.cur = kEmpty; T.sp = 0; T.cntNodes = 0; T.cntPrims = 0; T.cntSteps = 0; T.lit = true; T.shadow = mk3(1.0f, 1.0f, 1.0f);
	unsigned long long cn = 0, cp = 0;
	for (;;) {
		const uint32_t got = fetchWork(idx == kInactive, pool, &ctr->head_shadow, total);
		if (got != kInactive) {
			idx = got;
			const uint32_t s = idx % nsurf;
			const f3 p = mk3(qf(sq, SQ_PX, s), qf(sq, SQ_PY, s), qf(sq, SQ_PZ, s));
			f3 Ld, lightPos;
			pointLightGeometry(S.lights[lightList[idx / nsurf]], p, Ld, lightPos);
			travInit<true>(S, T, p, lightPos - p);   // bvh.cpp:124
		}
		if (__ballot(idx != kInactive) == 0) { if (workLeft(pool)) continue; break; }
		for (;;) {
			if (idx != kInactive) {
				if (T.cur != kEmpty) travRound<true, CSG, COUNT>(S, stack, T);
				if (T.cur == kEmpty) {
					if (COUNT) { cn += T.cntNodes; cp += T.cntPrims; }
					if (T.lit) {
						const rt_light L = S.lights[lightList[idx / nsurf]];
						const Surface sf = loadSurface(sq, idx % nsurf);
						f3 Ld, lightPos;
						pointLightGeometry(L, sf.p, Ld, lightPos);
						if (!(L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff))) {   // objects.cpp:451
							const f3 c = phong(ld3(L.colour) * T.shadow, sf.kd, sf.ks, sf.shininess, sf.N, Ld, sf.V);
							accumAdd(A, sf.pixKey / aaSq, sf.w * c);
						}
					}
					idx = kInactive;
				}
			}
			const int busy = __popcll(__ballot(idx != kInactive));
			if (busy == 0 || (workLeft(pool) && busy < kRefillBelow)) break;
		}
	}
	if (COUNT) { atomicAdd(&ctr->nodes_shadow, cn); atomicAdd(&ctr->prims_shadow, cp); }
}

// Packet form of k_shadow_point for coherent surface sets: 64 consecutive surfaces towards the same light.  Only the
// surface point is held during the walk; the rest of the surface record is read when the ray turned out lit.
template <bool CSG, bool COUNT>
__global__ __launch_bounds__(kBlock, RT_SHADOW_WAVES_PER_SIMD) void k_shadow_point_packet(DevScene S, RtQueue sq, uint32_t nsurfArg, const int32_t* lightList, uint32_t nLights,
                                                                RtAccum A, uint32_t aaSq, RtCounters* ctr, int countOnDevice, int oddCursor) {
	uint32_t* const head = oddCursor ? &ctr->head_shadow_odd : &ctr->head_shadow;
	// countOnDevice: the launch was queued before the host knew how many surfaces the shading kernel made (nsurfArg is only a bound)
	const uint32_t nsurf = countOnDevice ? (ctr->surf_count < sq.cap ? ctr->surf_count : sq.cap) : nsurfArg;
	const uint32_t lane = __lane_id();
	const PacketLds ws = packetLdsInit(S);
	const uint32_t packetsPerLight = (nsurf + 63u) / 64u;
	const uint32_t totalPackets = packetsPerLight * nLights;
	unsigned long long cn = 0, cp = 0, cs = 0;
	TR_DECL(1)
	// light-minor numbering: the packets of one group of 64 surfaces towards every light are neighbours
	auto surfaceOf = [&](uint32_t q) { return (q / nLights) * 64u + lane; };
	auto loadPoint = [&](uint32_t q) {
		f3 p = mk3(0, 0, 0);
		if (q != kInactive) {
			const uint32_t s = surfaceOf(q);
			if (s < nsurf) p = mk3(qf(sq, SQ_PX, s), qf(sq, SQ_PY, s), qf(sq, SQ_PZ, s));
		}
		return p;
	};
	PacketFeed feed = firstPackets(totalPackets);
	uint32_t q = fetchPacket(feed, head, totalPackets);
	while (q != kInactive) {
		TR(1, q);
		const f3 p = loadPoint(q);
		const int lightIndex = lightList[q % nLights];
		const uint32_t s = surfaceOf(q);
		const bool alive = s < nsurf;
		Trav T;
		T.cur = kEmpty; T.lit = true; T.shadow = mk3(1.0f, 1.0f, 1.0f); T.cntNodes = 0; T.cntPrims = 0; T.cntSteps = 0;
		if (alive) {
			f3 Ld, lightPos;
			pointLightGeometry(S.lights[lightIndex], p, Ld, lightPos);
			travInit<true>(S, T, p, lightPos - p);   // bvh.cpp:124
		}
		TR(2, __popcll(__ballot(T.cur != kEmpty)));
		packetTraverse<true, CSG, COUNT>(S, ws, T, alive TR_ARG);
		TR(9, 0);
		q = fetchPacket(feed, head, totalPackets);
		TR(10, 0);
		if (COUNT) { cn += T.cntNodes; cp += T.cntPrims; cs += T.cntSteps; }
		if (alive && T.lit) {
			const rt_light L = S.lights[lightIndex];
			const Surface sf = loadSurface(sq, s);
			f3 Ld, lightPos;
			pointLightGeometry(L, sf.p, Ld, lightPos);
			if (!(L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff))) {   // objects.cpp:451
				const f3 c = phong(ld3(L.colour) * T.shadow, sf.kd, sf.ks, sf.shininess, sf.N, Ld, sf.V);
				accumAdd(A, sf.pixKey / aaSq, sf.w * c);
			}
		}
		TR(11, 0);
		WL_PACKET(T, alive);
	}
	WL_END();
	if (COUNT) { atomicAdd(&ctr->nodes_shadow, cn); atomicAdd(&ctr->prims_shadow, cp); if (lane == 0) atomicAdd(&ctr->steps_shadow, cs); }
}

// Two lights per packet (packetTraverse2): lights lightList[2j] and lightList[2j + 1] for the same 64 surfaces; the surface
// record is read once and the two contributions reach the pixel as one atomic per channel.
template <bool CSG, bool COUNT, bool MULTI>
__global__ __launch_bounds__(kBlock, RT_SHADOW2_WAVES_PER_SIMD) void k_shadow_point_packet2(DevScene S, RtQueue sq, uint32_t nsurfArg, const int32_t* lightList, uint32_t nPairs,
                                                                 RtAccum A, uint32_t aaSq, RtCounters* ctr, int countOnDevice) {
	const uint32_t nsurf = countOnDevice ? (ctr->surf_count < sq.cap ? ctr->surf_count : sq.cap) : nsurfArg;
	const uint32_t lane = __lane_id();
	const PacketLds ws = packetLdsInit(S, true);
	const uint32_t totalPackets = ((nsurf + 63u) / 64u) * nPairs;
	unsigned long long cn = 0, cp = 0, cs = 0;
	PacketFeed feed = firstPackets(totalPackets);
	uint32_t q = fetchPacket(feed, &ctr->head_shadow, totalPackets);
	while (q != kInactive) {
		const uint32_t s = (q / nPairs) * 64u + lane;
		const int lightA = lightList[2u * (q % nPairs)], lightB = lightList[2u * (q % nPairs) + 1u];
		const bool alive = s < nsurf;
		f3 p = mk3(0, 0, 0);
		if (alive) p = mk3(qf(sq, SQ_PX, s), qf(sq, SQ_PY, s), qf(sq, SQ_PZ, s));
		Trav TA, TB;
		TA.cur = kEmpty; TA.lit = true; TA.shadow = mk3(1.0f, 1.0f, 1.0f); TA.cntNodes = 0; TA.cntPrims = 0; TA.cntSteps = 0;
		TB.cur = kEmpty; TB.lit = true; TB.shadow = mk3(1.0f, 1.0f, 1.0f); TB.cntNodes = 0; TB.cntPrims = 0; TB.cntSteps = 0;
		if (alive) {
			f3 Ld, lightPos;
			pointLightGeometry(S.lights[lightA], p, Ld, lightPos);
			travInit<true, false>(S, TA, p, lightPos - p);   // bvh.cpp:124
			pointLightGeometry(S.lights[lightB], p, Ld, lightPos);
			travInit<true, false>(S, TB, p, lightPos - p);
			coldSt3(ws.cold, COLD_DA, TA.d); coldSt3(ws.cold, COLD_DB, TB.d);
			ws.cold[COLD_LENA * 64] = TA.len; ws.cold[COLD_LENB * 64] = TB.len;
		}
		coldSt3(ws.cold, COLD_SHA, mk3(1.0f, 1.0f, 1.0f)); coldSt3(ws.cold, COLD_SHB, mk3(1.0f, 1.0f, 1.0f));
		packetTraverse2<CSG, COUNT>(S, ws, TA, TB, alive, alive);
		TA.shadow = coldLd3(ws.cold, COLD_SHA); TB.shadow = coldLd3(ws.cold, COLD_SHB);
		q = fetchPacket(feed, &ctr->head_shadow, totalPackets);
		if (COUNT) { cn += TA.cntNodes + TB.cntNodes; cp += TA.cntPrims + TB.cntPrims; cs += 2u * TA.cntSteps; }   // a step of the two-ray walk serves two rays per lane
		if constexpr (!MULTI) {
			if (alive && (TA.lit || TB.lit)) {
				const Surface sf = loadSurface(sq, s);
				f3 cA = mk3(0, 0, 0), cB = cA;
				bool hasA = false, hasB = false;
				if (TA.lit) {
					const rt_light L = S.lights[lightA];
					f3 Ld, lightPos;
					pointLightGeometry(L, sf.p, Ld, lightPos);
					if (!(L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff))) {   // objects.cpp:451
						cA = sf.w * phong(ld3(L.colour) * TA.shadow, sf.kd, sf.ks, sf.shininess, sf.N, Ld, sf.V);
						hasA = true;
					}
				}
				if (TB.lit) {
					const rt_light L = S.lights[lightB];
					f3 Ld, lightPos;
					pointLightGeometry(L, sf.p, Ld, lightPos);
					if (!(L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff))) {
						cB = sf.w * phong(ld3(L.colour) * TB.shadow, sf.kd, sf.ks, sf.shininess, sf.N, Ld, sf.V);
						hasB = true;
					}
				}
				accumAdd2(A, sf.pixKey / aaSq, cA, hasA, cB, hasB);
			}
		} else {
			f3 cA = mk3(0, 0, 0), cB = cA;
			bool hasA = false, hasB = false;
			uint32_t pixel = 0xffffffffu - lane;   // (MULTI: lanes with nothing to add are runs of their own)
			if (alive && (TA.lit || TB.lit)) {
				const Surface sf = loadSurface(sq, s);
				pixel = sf.pixKey / aaSq;
				if (TA.lit) {
					const rt_light L = S.lights[lightA];
					f3 Ld, lightPos;
					pointLightGeometry(L, sf.p, Ld, lightPos);
					if (!(L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff))) {   // objects.cpp:451
						cA = sf.w * phong(ld3(L.colour) * TA.shadow, sf.kd, sf.ks, sf.shininess, sf.N, Ld, sf.V);
						hasA = true;
					}
				}
				if (TB.lit) {
					const rt_light L = S.lights[lightB];
					f3 Ld, lightPos;
					pointLightGeometry(L, sf.p, Ld, lightPos);
					if (!(L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff))) {
						cB = sf.w * phong(ld3(L.colour) * TB.shadow, sf.kd, sf.ks, sf.shininess, sf.N, Ld, sf.V);
						hasB = true;
					}
				}
			}
			accumAddRows(A, pixel, cA, hasA, cB, hasB);   // (several samples per pixel: neighbouring lanes add up first)
		}
	}
	if (COUNT) { atomicAdd(&ctr->nodes_shadow, cn); atomicAdd(&ctr->prims_shadow, cp); if (lane == 0) atomicAdd(&ctr->steps_shadow, cs); }
}

// Sum of a value over the wave, delivered in lane 63: xor-butterflies inside each row of 16 lanes (quad_perm, row_half_mirror,
// row_mirror), then lane 15 of each row added into the next row and lane 31 into the upper half (row_bcast).  Additions pair
// neighbours, then neighbouring pairs, ...: the tree a Hillis-Steele scan builds for its last lane.
#define RT_DPP_ADD(v, ctrl, rows) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xf, false))
__device__ __forceinline__ float waveSumToLane63(float v) {
	RT_DPP_ADD(v, 0xB1, 0xf);    // quad_perm:[1,0,3,2]
	RT_DPP_ADD(v, 0x4E, 0xf);    // quad_perm:[2,3,0,1]
	RT_DPP_ADD(v, 0x141, 0xf);   // row_half_mirror
	RT_DPP_ADD(v, 0x140, 0xf);   // row_mirror
	RT_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15, into rows 1 and 3
	RT_DPP_ADD(v, 0x143, 0xc);   // row_bcast:31, into rows 2 and 3
	return v;
}
#undef RT_DPP_ADD

// Rectangular / circular area light `lightIndex`: one lane per (surface, sample), samples of one surface in
// consecutive lanes (their rays leave the same point towards neighbouring targets), summed across lanes by a
// segmented shuffle scan so each surface costs one atomic per wave instead of one per sample.
// WHOLE: the number of samples is a multiple of 64, so every wave is one surface.  DEFER: see packetTraverse
template <bool CSG, bool COUNT, bool WHOLE, bool DEFER>
__global__ __launch_bounds__(kBlock, RT_AREA_WAVES_PER_SIMD) void k_shadow_area(DevScene S, RtQueue sq, uint32_t nsurfArg, int lightIndex, uint32_t seed,
                                                        RtAccum A, uint32_t aaSq, RtCounters* ctr, int countOnDevice) {
	const uint32_t nsurf = countOnDevice ? deviceCount(nsurfArg, &ctr->surf_count) : nsurfArg;
	if ((unsigned long long)blockIdx.x * kBlock >= (unsigned long long)nsurf * (uint32_t)S.lights[lightIndex].samples) return;   // workgroup-uniform
	const PacketLds ws = packetLdsInit(S);
	const rt_light L = S.lights[lightIndex];
	const uint32_t ns = (uint32_t)L.samples;
	const unsigned long long j = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
	const bool live = j < (unsigned long long)nsurf * ns;
	uint32_t s = kInactive;
	f3 c = mk3(0, 0, 0), p = c;
	Trav T;
	T.cur = kEmpty; T.lit = true; T.shadow = mk3(1.0f, 1.0f, 1.0f); T.cntNodes = 0; T.cntPrims = 0; T.cntSteps = 0;
	f3 lightPos = c;
	if (live) {
		s = (uint32_t)(j / ns);
		const uint32_t smp = (uint32_t)(j % ns);
		// only the surface point is held during the walk; the rest of the record is read when the sample turned out lit
		p = mk3(qf(sq, SQ_PX, s), qf(sq, SQ_PY, s), qf(sq, SQ_PZ, s));
		float r1, r2;
		rng2(seed, qu(sq, SQ_PIX, s), qu(sq, SQ_PATH, s), (uint32_t)lightIndex, smp, r1, r2);   // stands in for RAND, arealight.cpp:5
		float xDisplacement, yDisplacement;
		if (L.type == RT_LIGHT_RECT) {   // arealight.cpp:37-42
			xDisplacement = (float)(((double)r1 - 0.5) * (double)L.width);
			yDisplacement = (float)(((double)r2 - 0.5) * (double)L.height);
		} else {   // arealight.cpp:79-87
			const float angle = (float)((double)(r1 * 2) * RTD_PI);
			const float displacement = L.radius * sqrtf(r2);
			xDisplacement = displacement * cosf(angle);
			yDisplacement = displacement * sinf(angle);
		}
		lightPos = ld3(L.position) + ld3(L.planeX) * xDisplacement + ld3(L.planeY) * yDisplacement;
		travInit<true>(S, T, p, lightPos - p);
	}
	packetTraverse<true, CSG, COUNT, DEFER>(S, ws, T, live);
	const uint32_t lane = __lane_id();
	if constexpr (WHOLE) {
		f3 w = mk3(0, 0, 0);
		uint32_t pixKey = 0;
		if (live) {
			if (COUNT) { atomicAdd(&ctr->nodes_shadow, T.cntNodes); atomicAdd(&ctr->prims_shadow, T.cntPrims); if ((threadIdx.x & 63) == 0) atomicAdd(&ctr->steps_shadow, (unsigned long long)T.cntSteps); }
			if (T.lit) {
				const Surface sf = loadSurface(sq, s);
				c = phong(ld3(L.colour) * T.shadow, sf.kd, sf.ks, sf.shininess, sf.N, normalize(lightPos - sf.p), sf.V);   // arealight.cpp:13-19
			}
		}
		uint32_t nextKey = s;
		if ((ns & 63u) == 0u) {
			// a multiple of 64 samples: the wave is one surface and lane 63 gets the sum through the DPP network instead of 24 LDS
			// permutes -- the same balanced tree of additions over the lanes in order as a scan would build
			c.x = waveSumToLane63(c.x); c.y = waveSumToLane63(c.y); c.z = waveSumToLane63(c.z);
		} else {
			// (never taken: rtk_shadow_area sends other sample counts to the variant below.  Kept because the register allocation of this
			// kernel, which runs at its 128-VGPR limit with spills, came out 10 % slower on C4 without it)
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const uint32_t ks2 = __shfl_up(s, off);
				const float ax = __shfl_up(c.x, off), ay = __shfl_up(c.y, off), az = __shfl_up(c.z, off);
				if (lane >= (uint32_t)off && ks2 == s) { c.x += ax; c.y += ay; c.z += az; }
			}
			nextKey = __shfl_down(s, 1);
		}
		if (live && (lane == 63u || nextKey != s)) {   // arealight.cpp:21
			w = mk3(qf(sq, SQ_WR, s), qf(sq, SQ_WG, s), qf(sq, SQ_WB, s));
			pixKey = qu(sq, SQ_PIX, s);
			accumAdd(A, pixKey / aaSq, w * (c / (float)ns));
		}
	} else {
		// Any other number of samples: a surface's samples straddle waves, and WHERE depends on the surface's slot in the queue, which
		// is whatever order the shading kernel's workgroups reached the append counter in -- float partial sums per wave would make the
		// last bit of a pixel differ from run to run.  Each sample's share goes to fixed point by itself instead and the lanes of a
		// pixel add up as integers (accumAddRows): the same bits whatever the order.
		f3 w = mk3(0, 0, 0);
		uint32_t pixKey = 0;
		bool lit = false;
		if (live) {
			if (COUNT) { atomicAdd(&ctr->nodes_shadow, T.cntNodes); atomicAdd(&ctr->prims_shadow, T.cntPrims); if ((threadIdx.x & 63) == 0) atomicAdd(&ctr->steps_shadow, (unsigned long long)T.cntSteps); }
			if (T.lit) {
				const Surface sf = loadSurface(sq, s);
				c = phong(ld3(L.colour) * T.shadow, sf.kd, sf.ks, sf.shininess, sf.N, normalize(lightPos - sf.p), sf.V);   // arealight.cpp:13-19
				w = sf.w; pixKey = sf.pixKey; lit = true;
			}
		}
		accumAddRows(A, lit ? pixKey / aaSq : 0xffffffffu - lane, w * (c / (float)ns), lit, mk3(0, 0, 0), false);
	}
}

// shadowRay() for caller-supplied queries (rt_shadow_batch)
template <bool CSG>
__global__ __launch_bounds__(kBlock) void k_shadow_query(DevScene S, const float* point, const float* lightPos, uint32_t n, uint8_t* lit, float* shadowOut) {
	__shared__ int32_t stack[kStack][kBlock];
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const f3 p = ld3(point + 3 * (size_t)i), l = ld3(lightPos + 3 * (size_t)i);
	Trav T;
	travInit<true>(S, T, p, l - p);   // bvh.cpp:124
	while (T.cur != kEmpty) travRound<true, CSG, false>(S, stack, T);
	if (lit) lit[i] = T.lit;
	if (shadowOut) { shadowOut[3 * (size_t)i] = T.shadow.x; shadowOut[3 * (size_t)i + 1] = T.shadow.y; shadowOut[3 * (size_t)i + 2] = T.shadow.z; }
}

// ---------------------------------------------------------------------------------------------------------
// texel state of re-hit textured triangles (RtTexFix, rt_kernels.h)
// ---------------------------------------------------------------------------------------------------------
// position of two rays of one ray tree in the order the reference executes them (depth first: a hit, its reflected ray's subtree,
// its transmitted ray's subtree): path 1 = the first ray, 2p = the reflection of p, 2p + 1 = its transmission
__device__ __forceinline__ bool pathBefore(uint32_t a, uint32_t b) {
	const int da = 31 - __clz((int)a), db = 31 - __clz((int)b);
	if (da > db) { const uint32_t up = a >> (da - db); return up != b && up < b; }   // b is an ancestor of a: b first
	const uint32_t up = b >> (db - da);
	if (up == a) return da != db;                                                     // a is an ancestor of b: a first
	return a < up;
}
// is ray q inside the subtree of the REFLECTED ray of the hit of ray p?
__device__ __forceinline__ bool inReflectionSubtree(uint32_t p, uint32_t q) {
	const int dp = 31 - __clz((int)p), dq = 31 - __clz((int)q);
	return dq > dp && (q >> (dq - dp - 1)) == 2u * p;
}

// One record per lane.  The lights are walked as Object::lightPoint walks them (objects.cpp:77-79), with the material the reference
// would have used (the texel of the last hit on the triangle inside the reflected ray's subtree) and with the material the shading
// kernel did use (the hit's own texel); the pixel gets the difference of the ambient and diffuse terms.  Shadow rays are those the
// shadow kernels traced (same origin, same targets, same samples of the area lights): they are not counted again.
template <bool CSG>
__global__ __launch_bounds__(kBlock) void k_texel_fix(DevScene S, RtTexFix tex, uint32_t count, RtAccum A, uint32_t aaSq, uint32_t seed) {
	__shared__ int32_t stack[kStack][kBlock];
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= count) return;
	const RtTexRec r = tex.rec[i];
	uint32_t best = kInactive, bestPath = 0;
	for (uint32_t j = tex.heads[r.pixKey / aaSq]; j != kInactive; j = tex.rec[j].next) {
		if (j == i) continue;
		const uint32_t q = tex.rec[j].path;
		if (tex.rec[j].pixKey != r.pixKey || tex.rec[j].tri != r.tri || !inReflectionSubtree(r.path, q)) continue;
		if (best == kInactive || pathBefore(bestPath, q)) { best = j; bestPath = q; }
	}
	if (best == kInactive) return;
	const float* c1 = tex.rec[best].c;
	if (c1[0] == r.c[0] && c1[1] == r.c[1] && c1[2] == r.c[2]) return;
	const f3 kNew = ld3(c1), kOld = ld3(r.c), p = ld3(r.p), N = ld3(r.N), wl = ld3(r.wl);
	auto diffuse = [](f3 I, f3 kd, float NL) {   // addDiffuse, raymath.cpp:34-42
		f3 dcol = I * kd * NL;
		if (dcol.x < 0) dcol.x = 0;
		if (dcol.y < 0) dcol.y = 0;
		if (dcol.z < 0) dcol.z = 0;
		return dcol;
	};
	f3 sumNew = mk3(0, 0, 0), sumOld = mk3(0, 0, 0);
	for (int l = 0; l < S.n_lights; l++) {
		const rt_light L = S.lights[l];
		if (L.type == RT_LIGHT_AMBIENT) { sumNew = sumNew + ld3(L.colour) * kNew; sumOld = sumOld + ld3(L.colour) * kOld; continue; }
		const bool area = L.type == RT_LIGHT_RECT || L.type == RT_LIGHT_CIRC;
		const uint32_t ns = area ? (uint32_t)L.samples : 1u;
		f3 totNew = mk3(0, 0, 0), totOld = mk3(0, 0, 0);
		for (uint32_t smp = 0; smp < ns; smp++) {
			f3 Ld, lightPos;
			if (!area) pointLightGeometry(L, p, Ld, lightPos);
			else {
				float r1, r2;
				rng2(seed, r.pixKey, r.path, (uint32_t)l, smp, r1, r2);
				float xDisplacement, yDisplacement;
				if (L.type == RT_LIGHT_RECT) {   // arealight.cpp:37-42
					xDisplacement = (float)(((double)r1 - 0.5) * (double)L.width);
					yDisplacement = (float)(((double)r2 - 0.5) * (double)L.height);
				} else {   // arealight.cpp:79-87
					const float angle = (float)((double)(r1 * 2) * RTD_PI);
					const float displacement = L.radius * sqrtf(r2);
					xDisplacement = displacement * cosf(angle);
					yDisplacement = displacement * sinf(angle);
				}
				lightPos = ld3(L.position) + ld3(L.planeX) * xDisplacement + ld3(L.planeY) * yDisplacement;
				Ld = normalize(lightPos - p);
			}
			Trav T;
			travInit<true>(S, T, p, lightPos - p);
			while (T.cur != kEmpty) travRound<true, CSG, false>(S, stack, T);
			if (!T.lit) continue;
			if (L.type == RT_LIGHT_SPOT && !((double)dot(Ld, -ld3(L.direction)) > L.cos_cutoff)) continue;   // objects.cpp:451
			const f3 I = ld3(L.colour) * T.shadow;
			const float NL = dot(N, Ld);
			totNew = totNew + diffuse(I, kNew, NL);
			totOld = totOld + diffuse(I, kOld, NL);
		}
		if (area) { totNew = totNew / (float)ns; totOld = totOld / (float)ns; }   // arealight.cpp:21
		sumNew = sumNew + totNew; sumOld = sumOld + totOld;
	}
	accumAdd(A, r.pixKey / aaSq, wl * sumNew);
	accumAdd(A, r.pixKey / aaSq, -(wl * sumOld));
}

// empties the lists the records hang in and the record counter: the next batch starts clean
__global__ __launch_bounds__(kBlock) void k_texel_reset(RtTexFix tex, uint32_t count, RtCounters* ctr, uint32_t aaSq) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i == 0) *tex.count = 0;
	if (i < count) tex.heads[tex.rec[i].pixKey / aaSq] = kInactive;
}

// fixed point -> float RGB for the pixels of local range [lp0, lp0+count); clears the accumulators for the next frame
__global__ __launch_bounds__(kBlock) void k_resolve(RtAccum A, int W, int H, int tileRank, int tileRanks, uint32_t lp0, uint32_t count, uint32_t aaSq,
                                                    int linear, float* rgb) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= count) return;
	uint32_t pixel;
	if (linear) pixel = lp0 + i;
	else {
		int x, y;
		if (!tilePixel(lp0 + i, W, H, tileRank, tileRanks, x, y)) return;
		pixel = (uint32_t)y * (uint32_t)W + (uint32_t)x;
	}
	float c[3];
	for (int k = 0; k < 3; k++) {
		unsigned long long* a = A.rgb + (size_t)k * A.npix + pixel;
		c[k] = (float)((double)(long long)*a * (1.0 / 4294967296.0));
		if (aaSq > 1) c[k] = c[k] / (float)aaSq;   // q1.cpp:140
		*a = 0ull;
	}
	if (A.nonfinite[pixel]) { c[0] = c[1] = c[2] = __int_as_float(0x7fc00000); A.nonfinite[pixel] = 0; }
	if (rgb) { rgb[3 * (size_t)pixel] = c[0]; rgb[3 * (size_t)pixel + 1] = c[1]; rgb[3 * (size_t)pixel + 2] = c[2]; }
}

// multi-GPU: dense packing of a rank's tiles for the gather, and the inverse on the root
__global__ __launch_bounds__(kBlock) void k_tiles_copy(const float* src, float* dst, int W, int H, int tileRank, int tileRanks, uint32_t count, int mode) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= count) return;
	int x, y;
	if (!tilePixel(i, W, H, tileRank, tileRanks, x, y)) return;
	const size_t pixel = (size_t)y * W + x;
	for (int k = 0; k < 3; k++) {
		if (mode == 0) dst[3 * (size_t)i + k] = src[3 * pixel + k];          // pack: frame -> dense
		else if (mode == 1) dst[3 * pixel + k] = src[3 * (size_t)i + k];     // unpack: dense -> frame
		else dst[3 * pixel + k] = src[3 * pixel + k];                        // merge: frame -> frame, owned pixels only
	}
}

// The same exchange with 8-bit pixels (3 B instead of 12 B per pixel over xGMI).  Quantised exactly as
// rt_quantize_rgb8 does on the host: NaN and negatives -> 0, above 1 -> 255, else c * 255 + 0.5 truncated.
//   mode 0  pack:   float frame (row 0 = bottom) -> dense bytes, 256 x 3 per owned tile, in tile order
//   mode 1  unpack: dense bytes -> 8-bit frame, row 0 = TOP (the BMP's order, as rt_quantize_rgb8 writes it)
//   mode 2  direct: float frame -> 8-bit frame, owned tiles only (the root's own share)
__device__ __forceinline__ uint8_t quantize8(float c) {
	if (!(c > 0.0f)) c = 0.0f;
	if (c > 1.0f) c = 1.0f;
	return (uint8_t)(c * 255.0f + 0.5f);
}
__global__ __launch_bounds__(kBlock) void k_tiles_rgb8(const void* src, uint8_t* dst, int W, int H, int tileRank, int tileRanks, uint32_t count, int mode) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= count) return;
	int x, y;
	const bool inside = tilePixel(i, W, H, tileRank, tileRanks, x, y);
	if (mode == 0) {   // padding slots of partial edge tiles are written too (zero): the payload has no undefined bytes
		const float* f = static_cast<const float*>(src);
		for (int k = 0; k < 3; k++) dst[3 * (size_t)i + k] = inside ? quantize8(f[3 * ((size_t)y * W + x) + k]) : (uint8_t)0;
		return;
	}
	if (!inside) return;
	const size_t out = (size_t)(H - 1 - y) * W + x;
	if (mode == 1) { const uint8_t* b = static_cast<const uint8_t*>(src); for (int k = 0; k < 3; k++) dst[3 * out + k] = b[3 * (size_t)i + k]; }
	else { const float* f = static_cast<const float*>(src); for (int k = 0; k < 3; k++) dst[3 * out + k] = quantize8(f[3 * ((size_t)y * W + x) + k]); }
}

// The counts of a level, written straight into host memory (pinned, mapped) for the host loop that runs ahead of the GPU: one
// 64-thread launch instead of a copy and an event in the stream, and the host sees them as soon as the store lands instead of
// after a copy engine and an interrupt.  `seq` (never 0) goes last, behind a system-scope fence: the host polls it.
__global__ __launch_bounds__(64) void k_publish(const RtCounters* ctr, const uint32_t* texCount, const uint32_t* stateCount, RtLevelReport* host, uint32_t seq) {
	if (threadIdx.x != 0) return;
	host->texrec_count = texCount ? *texCount : 0u;
	host->state_cases = stateCount ? *stateCount : 0u;
	host->next_count = ctr->next_count;
	host->surf_count = ctr->surf_count;
	host->overflow = ctr->overflow;
	__threadfence_system();
	__hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

inline unsigned gridFor(unsigned long long n) { return (unsigned)((n + kBlock - 1) / kBlock); }
// persistent kernels: enough workgroups to fill every CU once (the kernels loop over the work themselves)
inline unsigned gridPersistent(unsigned long long n, int maxBlocks) { return (unsigned)std::min<unsigned long long>((n + kBlock - 1) / kBlock, (unsigned long long)maxBlocks); }

}  // namespace

// =========================================================================================================
// launch wrappers (host)
// =========================================================================================================
// Every launch is checked on the spot and the FIRST failure of the calling thread is kept with the kernel's name
// (rtk_take_launch_error, read by the host loop where it next looks at errors): a bare hipGetLastError() at the next
// synchronisation point would say "invalid configuration" without saying whose.
static thread_local char g_launchError[320];
static void noteLaunch(const char* kernel) {
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess && !g_launchError[0]) snprintf(g_launchError, sizeof g_launchError, "launch of %s failed: %s", kernel, hipGetErrorString(e));
}
const char* rtk_take_launch_error() {
	static thread_local char taken[320];
	if (!g_launchError[0]) return nullptr;
	snprintf(taken, sizeof taken, "%s", g_launchError);
	g_launchError[0] = 0;
	return taken;
}
#define RT_LAUNCH_RAW(kernel, grid, block, lds, stream, ...) do { hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); noteLaunch(#kernel); } while (0)
#define RT_LAUNCH(kernel, n, stream, ...) do { if ((n) > 0) RT_LAUNCH_RAW(kernel, dim3(gridFor(n)), dim3(kBlock), 0, stream, __VA_ARGS__); } while (0)
// Persistent kernels start every wave on a statically numbered piece of work, so the grid must not exceed what is
// resident at once (a workgroup that only starts when another has finished would sit on its piece until the end).
static int computeUnits() {
	static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
	return cus;
}
// resident workgroups per CU of `kernel` with `lds` bytes of dynamic LDS (cached; launches come from several host threads)
template <class K> static int residentPerCU(K kernel, const char* name, size_t lds) {
	struct Entry { const void* k; size_t lds; int perCU; };
	static std::mutex m;
	static std::vector<Entry> cache;
	std::lock_guard<std::mutex> g(m);
	for (const Entry& e : cache) if (e.k == (const void*)kernel && e.lds == lds) return e.perCU;
	int perCU = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, kBlock, lds) != hipSuccess || perCU <= 0) perCU = 1;
	if (getenv("RT_DEBUG_GRID")) fprintf(stderr, "[grid] %s: %d workgroups/CU resident with %zu B LDS\n", name, perCU, lds);
	cache.push_back({ (const void*)kernel, lds, perCU });
	return perCU;
}
#define RT_LAUNCH_PS(kernel, n, maxBlocks, lds, stream, ...) do { if ((n) > 0) { \
		const int _perCU = residentPerCU(kernel, #kernel, (size_t)(lds)); \
		RT_LAUNCH_RAW(kernel, dim3(gridPersistent(n, std::min(maxBlocks, _perCU * computeUnits()))), dim3(kBlock), (size_t)(lds), stream, __VA_ARGS__); } } while (0)
#define RT_LAUNCH_P(kernel, n, maxBlocks, stream, ...) RT_LAUNCH_PS(kernel, n, maxBlocks, 0, stream, __VA_ARGS__)
#define RT_LAUNCH_S(kernel, n, lds, stream, ...) do { if ((n) > 0) RT_LAUNCH_RAW(kernel, dim3(gridFor(n)), dim3(kBlock), lds, stream, __VA_ARGS__); } while (0)
static size_t packetLdsBytes(const DevScene& S) { return (size_t)RT_PACKET_LDS_BYTES(S.quad_levels, S.treelet_staging); }

void rtk_generate(hipStream_t st, const rt_camera& cam, int W, int H, int aaX, int aaY, int tileRank, int tileRanks, uint32_t r0, uint32_t count, RtQueue rq) {
	RT_LAUNCH(k_generate, count, st, cam, W, H, aaX, aaY, tileRank, tileRanks, r0, count, rq);
}
void rtk_load_rays(hipStream_t st, const float* a, const float* b, uint32_t n, int bIsTarget, RtQueue rq, uint32_t pixBase) {
	RT_LAUNCH(k_load_rays, n, st, a, b, n, bIsTarget, rq, pixBase);
}
// a stride near the golden section of `groups` and coprime to it: consecutive hand-outs land far apart, every group is visited once
static uint32_t coprimeStride(uint32_t groups) {
	if (groups < 3) return 1;
	auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
	uint32_t s = (uint32_t)((double)groups * 0.6180339887498949);
	if (s < 1) s = 1;
	while (gcd(s, groups) != 1) s++;
	return s % groups ? s % groups : 1;
}
void rtk_intersect(hipStream_t st, const DevScene& S, bool csg, bool count, bool coherent, int maxBlocks, RtQueue rq, uint32_t n, RtQueue hq, RtCounters* ctr, const uint32_t* nDev) {
	if (coherent) {   // (the strided hand-out order needs the exact n: with a count on the device, n is a bound and the packets go out in order)
		const uint32_t stride = nDev ? 0u : coprimeStride(((n + 63u) / 64u + kPacketGroup - 1u) / kPacketGroup);
		if (csg) { if (count) RT_LAUNCH_PS((k_intersect_packet<true, true>), n, maxBlocks, packetLdsBytes(S), st, S, rq, n, nDev, hq, ctr, stride); else RT_LAUNCH_PS((k_intersect_packet<true, false>), n, maxBlocks, packetLdsBytes(S), st, S, rq, n, nDev, hq, ctr, stride); }
		else { if (count) RT_LAUNCH_PS((k_intersect_packet<false, true>), n, maxBlocks, packetLdsBytes(S), st, S, rq, n, nDev, hq, ctr, stride); else RT_LAUNCH_PS((k_intersect_packet<false, false>), n, maxBlocks, packetLdsBytes(S), st, S, rq, n, nDev, hq, ctr, stride); }
		return;
	}
	if (csg) { if (count) RT_LAUNCH_P((k_intersect<true, true>), n, maxBlocks, st, S, rq, n, nDev, hq, ctr); else RT_LAUNCH_P((k_intersect<true, false>), n, maxBlocks, st, S, rq, n, nDev, hq, ctr); }
	else { if (count) RT_LAUNCH_P((k_intersect<false, true>), n, maxBlocks, st, S, rq, n, nDev, hq, ctr); else RT_LAUNCH_P((k_intersect<false, false>), n, maxBlocks, st, S, rq, n, nDev, hq, ctr); }
}
void rtk_export_hits(hipStream_t st, const DevScene& S, RtQueue hq, uint32_t n, uint8_t* hit, int32_t* id, float* t) {
	RT_LAUNCH(k_export_hits, n, st, S, hq, n, hit, id, t);
}
void rtk_shade(hipStream_t st, const DevScene& S, bool csg, RtQueue rq, RtQueue hq, uint32_t n, RtQueue nq, RtQueue sq, RtQueue tq, RtCounters* ctr, RtAccum A,
               uint32_t aaSq, int maxDepth, int wantSurfaces, uint8_t* outHit, int32_t* outId, float* outT, RtPick pick, RtTexFix tex, const uint32_t* nDev) {
	if (pick.ev) {   // rt_trace_pick: the variants that narrate
		if (tex.rec) {
			if (csg) RT_LAUNCH((k_shade<true, true, true, false>), n, st, S, rq, hq, n, nDev, nq, sq, tq, ctr, A, aaSq, maxDepth, wantSurfaces, outHit, outId, outT, pick, tex);
			else RT_LAUNCH((k_shade<false, true, true, false>), n, st, S, rq, hq, n, nDev, nq, sq, tq, ctr, A, aaSq, maxDepth, wantSurfaces, outHit, outId, outT, pick, tex);
		} else if (csg) RT_LAUNCH((k_shade<true, true, false, false>), n, st, S, rq, hq, n, nDev, nq, sq, tq, ctr, A, aaSq, maxDepth, wantSurfaces, outHit, outId, outT, pick, tex);
		else RT_LAUNCH((k_shade<false, true, false, false>), n, st, S, rq, hq, n, nDev, nq, sq, tq, ctr, A, aaSq, maxDepth, wantSurfaces, outHit, outId, outT, pick, tex);
		return;
	}
#define RT_SHADE(...) RT_LAUNCH((k_shade<__VA_ARGS__>), n, st, S, rq, hq, n, nDev, nq, sq, tq, ctr, A, aaSq, maxDepth, wantSurfaces, outHit, outId, outT, pick, tex)
	// several samples per pixel: the variants that combine their accumulator atomics (the switch: for the tests, read per call).  From three
	// samples on: measured on C2 at 1080p with 2 / 4 / 8 samples per pixel (tools/dbg/aa_times.py), the row sums cost 3 % / gain 37 % / 56 %
	const bool multi = aaSq > 2 && getenv("RT_NO_ROW_SUMS") == nullptr;
	if (tex.rec) {   // scenes with reflective textured triangles
		if (csg) { if (multi) RT_SHADE(true, false, true, true); else RT_SHADE(true, false, true, false); }
		else { if (multi) RT_SHADE(false, false, true, true); else RT_SHADE(false, false, true, false); }
		return;
	}
	if (csg) { if (multi) RT_SHADE(true, false, false, true); else RT_SHADE(true, false, false, false); }
	else { if (multi) RT_SHADE(false, false, false, true); else RT_SHADE(false, false, false, false); }
#undef RT_SHADE
}
void rtk_level(hipStream_t st, const DevScene& S, const RtSmallScene& small, bool csg, bool inlineTransmit, RtQueue rq, uint32_t n, const uint32_t* nDev, RtQueue nq, RtQueue sq, RtQueue tq,
               RtCounters* ctr, RtAccum A, uint32_t aaSq, int maxDepth, int surfaces, const int32_t* lightList, uint32_t nLights, uint32_t* stateCases, RtLevelReport* report, uint32_t seq) {
#define RT_LEVEL_ARGS S, small, rq, n, nDev, nq, sq, tq, ctr, A, aaSq, maxDepth, surfaces, lightList, nLights, stateCases, report, seq
	if (small.blob && getenv("RT_NO_SMALL_SCENE") == nullptr) {   // (the switch: read per call, for the tests)
		const size_t lds = (size_t)small.stackRows * kBlock * 4u + small.bytes;
		if (csg && inlineTransmit) RT_LAUNCH_S((k_level<true, true, true>), n, lds, st, RT_LEVEL_ARGS);
		else if (csg) RT_LAUNCH_S((k_level<true, true, false>), n, lds, st, RT_LEVEL_ARGS);
		else if (inlineTransmit) RT_LAUNCH_S((k_level<false, true, true>), n, lds, st, RT_LEVEL_ARGS);
		else RT_LAUNCH_S((k_level<false, true, false>), n, lds, st, RT_LEVEL_ARGS);
		return;
	}
	if (csg && inlineTransmit) RT_LAUNCH((k_level<true, false, true>), n, st, RT_LEVEL_ARGS);
	else if (csg) RT_LAUNCH((k_level<true, false, false>), n, st, RT_LEVEL_ARGS);
	else if (inlineTransmit) RT_LAUNCH((k_level<false, false, true>), n, st, RT_LEVEL_ARGS);
	else RT_LAUNCH((k_level<false, false, false>), n, st, RT_LEVEL_ARGS);
#undef RT_LEVEL_ARGS
}
void rtk_texel_fix(hipStream_t st, const DevScene& S, bool csg, RtTexFix tex, uint32_t count, RtCounters* ctr, RtAccum A, uint32_t aaSq, uint32_t seed, bool apply) {
	if (!tex.rec || count == 0) return;
	count = count < tex.cap ? count : tex.cap;
	if (apply) {
		if (csg) RT_LAUNCH((k_texel_fix<true>), count, st, S, tex, count, A, aaSq, seed);
		else RT_LAUNCH((k_texel_fix<false>), count, st, S, tex, count, A, aaSq, seed);
	}
	RT_LAUNCH(k_texel_reset, count, st, tex, count, ctr, aaSq);
}
void rtk_transmit(hipStream_t st, const DevScene& S, bool csg, RtQueue tq, uint32_t maxRequests, RtQueue nq, RtCounters* ctr, uint32_t* stateCases, RtAccum A, uint32_t aaSq, int maxDepth, RtPick pick) {
	if (maxRequests == 0) return;
	const dim3 grid((maxRequests + kTransmitBlock - 1) / kTransmitBlock), block(kTransmitBlock);
	const size_t lds = RTD_CSG_LDS_BYTES(kTransmitBlock, S.csg_cap, S.csg_nodes_max);
	static const bool noLds = getenv("RT_NO_CSG_LDS") != nullptr;
	if (pick.ev) {   // rt_trace_pick: one ray tree, no need for the LDS lists
		if (csg) RT_LAUNCH_RAW((k_transmit<true, false, true>), grid, block, 0, st, S, tq, nq, ctr, stateCases, A, aaSq, maxDepth, pick);
		else RT_LAUNCH_RAW((k_transmit<false, false, true>), grid, block, 0, st, S, tq, nq, ctr, stateCases, A, aaSq, maxDepth, pick);
		return;
	}
	if (!csg) RT_LAUNCH_RAW((k_transmit<false, false, false>), grid, block, 0, st, S, tq, nq, ctr, stateCases, A, aaSq, maxDepth, pick);
	else if (lds <= 64u * 1024u && !noLds) RT_LAUNCH_RAW((k_transmit<true, true, false>), grid, block, lds, st, S, tq, nq, ctr, stateCases, A, aaSq, maxDepth, pick);
	else RT_LAUNCH_RAW((k_transmit<true, false, false>), grid, block, 0, st, S, tq, nq, ctr, stateCases, A, aaSq, maxDepth, pick);
}
void rtk_shadow_point(hipStream_t st, const DevScene& S, bool csg, bool count, bool coherent, int maxBlocks, RtQueue sq, uint32_t nsurf, const int32_t* lightList, uint32_t nLights,
                      RtAccum A, uint32_t aaSq, RtCounters* ctr, bool countOnDevice, int maxBlocksPairs) {
	const int cod = countOnDevice ? 1 : 0;   // packet kernels only: nsurf is a bound, the kernels read the count from ctr
	const unsigned long long items = (unsigned long long)nsurf * nLights;
	const bool noPairs = getenv("RT_NO_LIGHT_PAIRS") != nullptr;   // (read per call: the tests switch it)
	if (coherent && nLights >= 2 && !noPairs) {   // two lights per packet; an odd last one in a launch of its own (with its own cursor)
		const uint32_t nPairs = nLights / 2;
		const unsigned long long pitems = (unsigned long long)((nsurf + 63u) / 64u) * 64u * nPairs;
		if (nLights % 2) {
			const unsigned long long oitems = (unsigned long long)((nsurf + 63u) / 64u) * 64u;
			const int32_t* last = lightList + (nLights - 1);
			if (csg) { if (count) RT_LAUNCH_PS((k_shadow_point_packet<true, true>), oitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, last, 1u, A, aaSq, ctr, cod, 1); else RT_LAUNCH_PS((k_shadow_point_packet<true, false>), oitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, last, 1u, A, aaSq, ctr, cod, 1); }
			else { if (count) RT_LAUNCH_PS((k_shadow_point_packet<false, true>), oitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, last, 1u, A, aaSq, ctr, cod, 1); else RT_LAUNCH_PS((k_shadow_point_packet<false, false>), oitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, last, 1u, A, aaSq, ctr, cod, 1); }
		}
#define RT_PAIRS(...) RT_LAUNCH_PS((k_shadow_point_packet2<__VA_ARGS__>), pitems, maxBlocksPairs > 0 ? maxBlocksPairs : maxBlocks, packetLdsBytes(S) + (RT_BLOCK / 64) * RT_PACKET_COLD_BYTES, st, S, sq, nsurf, lightList, nPairs, A, aaSq, ctr, cod)
		if (aaSq > 2 && getenv("RT_NO_ROW_SUMS") == nullptr) {   // several samples per pixel: the variants that combine their accumulator atomics (accumAddRows)
			if (csg) { if (count) RT_PAIRS(true, true, true); else RT_PAIRS(true, false, true); }
			else { if (count) RT_PAIRS(false, true, true); else RT_PAIRS(false, false, true); }
		} else {
			if (csg) { if (count) RT_PAIRS(true, true, false); else RT_PAIRS(true, false, false); }
			else { if (count) RT_PAIRS(false, true, false); else RT_PAIRS(false, false, false); }
		}
#undef RT_PAIRS
		return;
	}
	if (coherent) {
		const unsigned long long pitems = (unsigned long long)((nsurf + 63u) / 64u) * 64u * nLights;
		if (csg) { if (count) RT_LAUNCH_PS((k_shadow_point_packet<true, true>), pitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod, 0); else RT_LAUNCH_PS((k_shadow_point_packet<true, false>), pitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod, 0); }
		else { if (count) RT_LAUNCH_PS((k_shadow_point_packet<false, true>), pitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod, 0); else RT_LAUNCH_PS((k_shadow_point_packet<false, false>), pitems, maxBlocks, packetLdsBytes(S), st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod, 0); }
		return;
	}
	if (csg) { if (count) RT_LAUNCH_P((k_shadow_point<true, true>), items, maxBlocks, st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod); else RT_LAUNCH_P((k_shadow_point<true, false>), items, maxBlocks, st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod); }
	else { if (count) RT_LAUNCH_P((k_shadow_point<false, true>), items, maxBlocks, st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod); else RT_LAUNCH_P((k_shadow_point<false, false>), items, maxBlocks, st, S, sq, nsurf, lightList, nLights, A, aaSq, ctr, cod); }
}
void rtk_shadow_area(hipStream_t st, const DevScene& S, bool csg, int nCsg, bool count, RtQueue sq, uint32_t nsurf, int lightIndex, uint32_t samples, uint32_t seed,
                     RtAccum A, uint32_t aaSq, RtCounters* ctr, bool countOnDevice) {
	const int cod = countOnDevice ? 1 : 0;
	const unsigned long long items = (unsigned long long)nsurf * samples;
#define RT_AREA(...) RT_LAUNCH_S((k_shadow_area<__VA_ARGS__>), items, packetLdsBytes(S), st, S, sq, nsurf, lightIndex, seed, A, aaSq, ctr, cod)
	const bool defer = csg && RT_PACKET_DEFER_FITS(nCsg) && getenv("RT_NO_CSG_DEFER") == nullptr;   // (the switch: for the tests, read per call)
	if (samples % 64u == 0u) {   // every wave is one surface
		if (defer) { if (count) RT_AREA(true, true, true, true); else RT_AREA(true, false, true, true); }
		else if (csg) { if (count) RT_AREA(true, true, true, false); else RT_AREA(true, false, true, false); }
		else { if (count) RT_AREA(false, true, true, false); else RT_AREA(false, false, true, false); }
	} else {
		if (defer) { if (count) RT_AREA(true, true, false, true); else RT_AREA(true, false, false, true); }
		else if (csg) { if (count) RT_AREA(true, true, false, false); else RT_AREA(true, false, false, false); }
		else { if (count) RT_AREA(false, true, false, false); else RT_AREA(false, false, false, false); }
	}
#undef RT_AREA
}
void rtk_shadow_query(hipStream_t st, const DevScene& S, bool csg, const float* point, const float* lightPos, uint32_t n, uint8_t* lit, float* shadow) {
	if (csg) RT_LAUNCH((k_shadow_query<true>), n, st, S, point, lightPos, n, lit, shadow);
	else RT_LAUNCH((k_shadow_query<false>), n, st, S, point, lightPos, n, lit, shadow);
}
void rtk_resolve(hipStream_t st, RtAccum A, int W, int H, int tileRank, int tileRanks, uint32_t lp0, uint32_t count, uint32_t aaSq, int linear, float* rgb) {
	RT_LAUNCH(k_resolve, count, st, A, W, H, tileRank, tileRanks, lp0, count, aaSq, linear, rgb);
}
void rtk_tiles_copy(hipStream_t st, const float* src, float* dst, int W, int H, int tileRank, int tileRanks, uint32_t count, int mode) {
	RT_LAUNCH(k_tiles_copy, count, st, src, dst, W, H, tileRank, tileRanks, count, mode);
}
void rtk_tiles_rgb8(hipStream_t st, const void* src, uint8_t* dst, int W, int H, int tileRank, int tileRanks, uint32_t count, int mode) {
	RT_LAUNCH(k_tiles_rgb8, count, st, src, dst, W, H, tileRank, tileRanks, count, mode);
}
void rtk_publish(hipStream_t st, const RtCounters* ctr, const uint32_t* texCount, const uint32_t* stateCount, RtLevelReport* hostMapped, uint32_t seq) {
	RT_LAUNCH_RAW(k_publish, dim3(1), dim3(64), 0, st, ctr, texCount, stateCount, hostMapped, seq);
}
