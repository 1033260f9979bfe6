// rt_device.h -- device-side geometry and shading primitives for gfx950 (included only by rt_kernels.hip).
//
// Everything here is the per-lane arithmetic of the reference's hot path, re-expressed for a 64-wide wavefront:
// no virtual dispatch, no per-object hit caches (the hit point is recomputed from (e, d, t) -- every rayhit() in
// the reference sets cachedHitpoint = e + t*d), no heap.  Results must equal the reference's bit for bit wherever
// they decide a hit, so the float/double mix and GLM's operation order are kept and the file is compiled with
// -ffp-contract=off (no FMA contraction; hipcc's default correctly-rounded fp32 divide/sqrt).
// Reference citations are relative to /root/reference/src.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_scene.h"
#include "csg_sort.h"
#include "glibc_math.h"

namespace rtd {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// glm/detail/func_geometric.inl:54-60, 74-85, 14-19, 88-96
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
__device__ __forceinline__ float length(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 normalize(f3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
__device__ __forceinline__ bool isZero3(const float* v) { return v[0] == 0 && v[1] == 0 && v[2] == 0; }   // raymath.cpp:59-61

#define RTD_PI 3.14159265358979323846264338327950288   /* objects.h:9 */

// ---- counter-based replacement for rand() (same function as oracle/restate.cpp rng2) -------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ void rng2(uint32_t seed, uint32_t ray, uint32_t path, uint32_t light, uint32_t sample, float& u1, float& u2) {
	uint32_t h = mix32(seed ^ 0x9e3779b9u);
	h = mix32(h ^ ray);
	h = mix32(h + path * 0x85ebca6bu);
	h = mix32(h ^ (light * 0xc2b2ae35u + sample));
	uint32_t a = mix32(h ^ 0x27d4eb2fu), b = mix32(h + 0x165667b1u);
	u1 = (float)(a >> 8) * (1.0f / 16777216.0f);
	u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
}

// ---- BoundingBox::intersect, objects.cpp:11-44 --------------------------------------------------------
// The reference leaves the axis loop early with -1; tnear only grows, tfar only shrinks and neither ever becomes
// NaN (they are assigned under an ordered compare), so "tnear > tfar || tfar < 0" stays true once true and the
// three axes can be evaluated unconditionally with the same outcome.  Returns -1 (miss), else tnear/tfar clamped
// at 0 exactly as the reference does.
__device__ __forceinline__ void slab(float mn, float mx, float e, float d, float& tnear, float& tfar, bool& miss) {
	miss = miss || (d == 0 && (e < mn || e > mx));
	float t1 = (mn - e) / d;
	float t2 = (mx - e) / d;
	if (t1 > t2) { float s = t1; t1 = t2; t2 = s; }
	if (t1 > tnear) tnear = t1;
	if (t2 < tfar) tfar = t2;
	miss = miss || (tnear > tfar) || (tfar < 0);
}
__device__ __forceinline__ float boxIntersect(const float* b, f3 e, f3 d, bool exitHit) {
	float tnear = -RT_MAX_T, tfar = RT_MAX_T;
	bool miss = false;
	slab(b[0], b[1], e.x, d.x, tnear, tfar, miss);
	slab(b[2], b[3], e.y, d.y, tnear, tfar, miss);
	slab(b[4], b[5], e.z, d.z, tnear, tfar, miss);
	if (miss) return -1.0f;
	if (!exitHit) return tnear < 0 ? 0.0f : tnear;
	return tfar < 0 ? 0.0f : tfar;
}

// ---- Sphere::rayhit, objects.cpp:159-178 (fp64 island kept op for op; pow(x,2) in double is exactly x*x) ----
__device__ __forceinline__ float sphereHit(f3 center, float radius, f3 e, f3 d, bool exitHit) {
	f3 ec = e - center;
	float dde = dot(d, ec);
	float dd = dot(d, d);
	double disc = (double)dde * (double)dde - (double)(dd * (dot(ec, ec) - radius * radius));
	if (disc < 0) return 0.0f;
	double rest = (double)(dot(-d, ec) / dd);
	double q = sqrt(disc) / (double)dd;
	double t = exitHit ? rest + q : rest - q;
	if (t < 0) return 0.0f;
	return (float)t;
}

// ---- Plane::rayhit, objects.cpp:199-213 ----
__device__ __forceinline__ float planeHit(f3 point, f3 normal, f3 e, f3 d, bool exitHit) {
	f3 n = exitHit ? -normal : normal;
	double numerator = (double)dot(n, point - e);
	double denominator = (double)dot(n, d);
	double t = numerator / denominator;
	if (t <= 0 || numerator > 0) return 0.0f;
	return (float)t;
}

// ---- Triangle::rayhit objects.cpp:251-261 + pointInTriangle raymath.cpp:26-32 ----
__device__ __forceinline__ float triHit(f3 p0, f3 p1, f3 p2, f3 n, f3 e, f3 d, bool exitHit) {
	float t = planeHit(p0, n, e, d, exitHit);
	f3 hitpos = e + t * d;
	float test1 = dot(cross(hitpos - p0, p1 - p0), n);
	float test2 = dot(cross(hitpos - p1, p2 - p1), n);
	float test3 = dot(cross(hitpos - p2, p0 - p2), n);
	bool inside = (test1 >= 0 && test2 >= 0 && test3 >= 0) || (test1 <= 0 && test2 <= 0 && test3 <= 0);
	if (t <= 0 || !inside) return 0.0f;
	return t;
}
// What Plane::rayhit's early return (objects.cpp:206) decides from signs alone, without the divide: with numerator = n.(p0 - e)
// and denominator = n.d as the reference computes them, t = numerator / denominator is positive with numerator <= 0
// only if both are negative (zero, NaN and -0 denominators all end in a miss further down).  A shadow query also needs
// t < 1 to count (bvh.cpp:148), and for negative floats that is exactly numerator > denominator: the quotient of two
// distinct floats is at most 1 - 2^-24 and rounds below 1.  False = the triangle cannot be a hit for this ray.
template <bool ANYHIT>
__device__ __forceinline__ bool triCandidate(f3 p0, f3 n, f3 e, f3 d) {
	const float num = dot(n, p0 - e), den = dot(n, d);
	return num < 0.0f && den < 0.0f && (!ANYHIT || num > den);
}
__device__ __forceinline__ float triHit(const rt_tri* tr, f3 e, f3 d, bool exitHit) {
	// 64 B record: four 16 B loads
	const float4* q = reinterpret_cast<const float4*>(tr);
	float4 a = q[0], b = q[1], c = q[2];
	return triHit(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), mk3(c.y, c.z, c.w), e, d, exitHit);
}

// ---- Box::rayhit objects.cpp:337-346 / Box::getNormal :348-361 ----
__device__ __forceinline__ float boxHit(const float* b, f3 e, f3 d, bool exitHit) {
	float t = boxIntersect(b, e, d, exitHit);
	return t < 0 ? 0.0f : t;
}
__device__ __forceinline__ void boxNormal(const float* b, f3 hp, f3& n) {   // n is left untouched when no face matches
	if ((double)fabsf(hp.x - b[0]) < 1e-5) n = mk3(-1, 0, 0);
	else if ((double)fabsf(hp.x - b[1]) < 1e-5) n = mk3(1, 0, 0);
	else if ((double)fabsf(hp.y - b[2]) < 1e-5) n = mk3(0, -1, 0);
	else if ((double)fabsf(hp.y - b[3]) < 1e-5) n = mk3(0, 1, 0);
	else if ((double)fabsf(hp.z - b[4]) < 1e-5) n = mk3(0, 0, -1);
	else if ((double)fabsf(hp.z - b[5]) < 1e-5) n = mk3(0, 0, 1);
}

// Everything a kernel needs to know about the scene; passed by value (kernel arguments), pointers are HBM.
#define RT_MESH_STACK 40   /* meshHit's stack: depth of a median-split tree over a mesh's triangles (checked at upload) */
struct DevScene {
	const rt_plane* planes;
	const rt_sphere* spheres;
	const rt_tri* tris;
	const rt_box* boxes;
	const rt_csg* csgs;
	const rt_csg_node* csg_nodes;
	const rt_mesh* meshes;
	const rt_tri* mesh_tris;
	const rt_uv* uvs;
	const rt_material* materials;
	const rt_light* lights;
	const rt_texture* textures;
	const uint8_t* texels;
	const uint32_t* leaf_prims;
	const rt_bvh_node* mesh_nodes;     // per-mesh trees for Mesh::rayhit (meshHit); device-only, built at upload
	const int32_t* mesh_order;         // their leaves: offsets into a mesh's triangles
	const int32_t* mesh_root;          // per mesh: root node, or -1 = scan the triangles
	const float4* treelets;            // the same tree cut into 1 KB treelets of up to 15 nodes (rt_kernels.h), device-only
	const float4* quads;               // ... and once more as quad nodes (two slots each) for the packet walks (rt_kernels.h), device-only
	int32_t quad_levels;               // treelets on the longest root-to-leaf path of that form (sizes the packet kernels' LDS)
	int32_t treelet_root;              // 0 (slot 0 of treelet 0) or, for a one-leaf tree, the leaf reference
	uint32_t treelet_root_aux;
	int32_t treelet_levels;            // treelets on the longest root-to-leaf path (sizes the packet kernels' LDS)
	int32_t treelet_staging;           // some leaf is an external run (needs the staging buffer)
	int32_t csg_cap;                   // intervals the largest CSG program of the scene can hold at any step (<= RTD_CSG_CAP)
	int32_t csg_nodes_max;             // nodes of the longest CSG program
	int32_t n_planes, n_lights;
	int32_t root_child, root_count;
	float root_box[6];
	float background[3];
};

// ---- Mesh::rayhit, objects.cpp:286-304 (linear scan in scene order; first of equal t wins) ----
// The result is a pure function of the triangles' own hit tests (the nearest positive t, lowest index among equals), so
// for a mesh of some size the scan is replaced by a walk of a tree over its triangles whose boxes are conservative
// (inflated at build time; a test too close to call counts as a hit): it skips only triangles that cannot be hit
// nearer than the best so far.  SURVEY.md section 8f-3: without it a refractive mesh costs O(triangles) per bounce.
__device__ __forceinline__ bool meshBoxMayHit(const float* b, f3 e, f3 inv, float tlimit) {
	const float x0 = (b[0] - e.x) * inv.x, x1 = (b[1] - e.x) * inv.x;
	const float y0 = (b[2] - e.y) * inv.y, y1 = (b[3] - e.y) * inv.y;
	const float z0 = (b[4] - e.z) * inv.z, z1 = (b[5] - e.z) * inv.z;
	const float lo = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
	const float hi = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
	const float u = fminf(hi, tlimit);
	const float m = 1e-6f * (fabsf(lo) + fabsf(u)) + 1e-30f;
	return !(u - lo < -m || hi < -1e-30f);   // not certainly missed / behind / beyond the best hit (NaN: visit)
}
__device__ inline float meshHit(const DevScene& S, int meshIndex, f3 e, f3 d, bool exitHit, f3& nrm) {
	const rt_mesh m = S.meshes[meshIndex];
	const rt_tri* tris = S.mesh_tris + m.tri_first;
	float t_min = RT_MAX_T;
	const int32_t root = S.mesh_root[meshIndex];
	const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	const bool finite = fabsf(inv.x) <= 3.0e38f && fabsf(inv.y) <= 3.0e38f && fabsf(inv.z) <= 3.0e38f;
	if (root < 0 || !finite) {   // small mesh, or a ray parallel to an axis (0 * inf in the slab test): the reference's scan
		for (int i = 0; i < m.tri_count; i++) {
			float t = triHit(tris + i, e, d, exitHit);
			if (t > 0.0f && t < t_min) { nrm = ld3(tris[i].n); t_min = t; }
		}
		return t_min == RT_MAX_T ? 0.0f : t_min;
	}
	int32_t stack[RT_MESH_STACK];
	int sp = 0, best = 0x7fffffff;
	int32_t cur = root;
	for (;;) {
		const rt_bvh_node* n = S.mesh_nodes + cur;
		int32_t next = -1;
		for (int k = 0; k < 2; k++) {
			if (!meshBoxMayHit(n->box[k], e, inv, t_min)) continue;
			const int32_t c = n->child[k];
			if (c >= 0) { if (next < 0) next = c; else if (sp < RT_MESH_STACK) stack[sp++] = c; continue; }
			const int32_t* order = S.mesh_order + ~c;
			for (int j = 0; j < n->count[k]; j++) {
				const int i = order[j];
				const float t = triHit(tris + i, e, d, exitHit);
				if (t > 0.0f && (t < t_min || (t == t_min && i < best))) { nrm = ld3(tris[i].n); t_min = t; best = i; }
			}
		}
		if (next >= 0) cur = next;
		else if (sp > 0) cur = stack[--sp];
		else break;
	}
	return t_min == RT_MAX_T ? 0.0f : t_min;
}

// ---- CSG: csg_node::setIntervals csg.cpp:95-232, csgObject::rayhit csg.cpp:8-26 ------------------------------
// An interval boundary carries its t and, instead of the normal itself, a TAG saying where the normal comes from:
// bits 0-7 node index inside the object's program, bit 8 = the leaf's far (exit) hit, bit 9 = negated an odd
// number of times by Difference (csg.cpp:211-212).  The winning boundary's normal is recomputed from its tag, which
// repeats the leaf's own computation and is therefore identical.
struct CsgB { float t; uint32_t tag; };
struct CsgIv { CsgB a, b; };
#define RTD_CSG_CAP (2 * RT_MAX_CSG_LEAVES)   /* intervals alive over the whole evaluation stack */

__device__ __forceinline__ CsgB csgMaxB(CsgB x, CsgB y) { return (x.t < y.t) ? y : x; }   // std::max(a,b,cmp): a unless a<b
__device__ __forceinline__ CsgB csgMinB(CsgB x, CsgB y) { return (y.t < x.t) ? y : x; }   // std::min(a,b,cmp): a unless b<a

__device__ inline float csgLeafHit(const DevScene& S, const rt_csg_node& nd, f3 e, f3 d, bool exitHit, f3* nrm) {
	if (nd.kind == RT_CSG_SPHERE) {
		f3 c = mk3(nd.f[0], nd.f[1], nd.f[2]);
		float t = sphereHit(c, nd.f[3], e, d, exitHit);
		if (nrm) *nrm = normalize((e + t * d) - c);   // Sphere::getNormal, objects.cpp:180-182 (only used when t > 0)
		return t;
	}
	if (nd.kind == RT_CSG_BOX) {
		float t = boxHit(nd.f, e, d, exitHit);
		if (nrm) { f3 n = mk3(0, 0, 0); boxNormal(nd.f, e + t * d, n); *nrm = n; }
		return t;
	}
	f3 n = mk3(0, 0, 0);
	float t = meshHit(S, nd.mesh, e, d, exitHit, n);
	if (nrm) *nrm = n;
	return t;
}

// Working storage of one evaluation: the interval lists of the evaluation stack (`arena`), the list an operator is
// producing (`tmp`).  CsgScratch keeps them in the lane's private (scratch) memory at
// the full capacity; CsgLds in a slice of the workgroup's LDS sized for the scene's largest program (DevScene::csg_cap),
// element i of thread t at [i * threads + t] -- an evaluation is a chain of dependent reads and writes of these lists,
// and LDS answers in ~100 cycles where scratch memory takes an L2 round trip.
// start[k] = arena offset of list k of the evaluation stack (start[sp] = end of the last one): 16 values of at most
// RTD_CSG_CAP, kept as bytes of two registers -- every step of a program begins by reading two or three of them, and as
// an indexed array they would sit in memory at the head of each step's dependency chain.  start[0] is always 0.
struct CsgStarts {
	unsigned long long lo = 0, hi = 0;
	__device__ __forceinline__ int get(int i) const {
		const int j = i - 1;
		const unsigned long long w = j < 8 ? lo : hi;
		return i == 0 ? 0 : (int)((w >> ((j & 7) * 8)) & 0xffull);
	}
	__device__ __forceinline__ void set(int i, int v) {   // i >= 1
		const int j = i - 1, sh = (j & 7) * 8;
		const unsigned long long m = ~(0xffull << sh), b = (unsigned long long)(uint32_t)v << sh;
		if (j < 8) lo = (lo & m) | b; else hi = (hi & m) | b;
	}
};
static_assert(RT_MAX_CSG_LEAVES <= 16 && 2 * RT_MAX_CSG_LEAVES <= 255, "CsgStarts holds 16 one-byte offsets");

struct CsgScratch {
	CsgIv arena_[RTD_CSG_CAP], tmp_[RTD_CSG_CAP];
	__device__ __forceinline__ int cap() const { return RTD_CSG_CAP; }
	__device__ __forceinline__ CsgIv& arena(int i) { return arena_[i]; }
	__device__ __forceinline__ CsgIv& tmp(int i) { return tmp_[i]; }
	__device__ __forceinline__ void loadProgram(const DevScene&, const rt_csg&) {}
	__device__ __forceinline__ rt_csg_node node(const DevScene& S, const rt_csg& c, int k) const { return S.csg_nodes[c.node_first + k]; }
};
// ... and a copy of the object's program, fetched with all its loads in flight at once (loadProgram) instead of one
// L2 round trip per node in the middle of the evaluation.
struct CsgLds {
	CsgIv* arena_; CsgIv* tmp_; float4* nodes_;   // this thread's element 0 of each
	int threads, cap_;
	__device__ __forceinline__ int cap() const { return cap_; }
	__device__ __forceinline__ CsgIv& arena(int i) { return arena_[i * threads]; }
	__device__ __forceinline__ CsgIv& tmp(int i) { return tmp_[i * threads]; }
	__device__ __forceinline__ void loadProgram(const DevScene& S, const rt_csg& c) {
		const float4* src = reinterpret_cast<const float4*>(S.csg_nodes + c.node_first);
		for (int k = 0; k < c.node_count; k += 4) {
			float4 v[8];
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = src[2 * k + (2 * k + j < 2 * c.node_count ? j : 0)];
#pragma unroll
			for (int j = 0; j < 8; j++) if (2 * k + j < 2 * c.node_count) nodes_[(2 * k + j) * threads] = v[j];
		}
	}
	__device__ __forceinline__ rt_csg_node node(const DevScene&, const rt_csg&, int k) const {
		const float4 a = nodes_[2 * k * threads], b = nodes_[(2 * k + 1) * threads];
		rt_csg_node nd;
		nd.kind = __float_as_int(a.x); nd.mesh = __float_as_int(a.y);
		nd.f[0] = a.z; nd.f[1] = a.w; nd.f[2] = b.x; nd.f[3] = b.y; nd.f[4] = b.z; nd.f[5] = b.w;
		return nd;
	}
};
#define RTD_CSG_LDS_BYTES(threads, cap, nodes) ((size_t)(threads) * (2u * (size_t)(cap) * sizeof(CsgIv) + (size_t)(nodes) * sizeof(rt_csg_node)))
__device__ __forceinline__ CsgLds csgLdsStore(void* lds, int threads, int thread, int cap) {
	CsgLds st;
	st.arena_ = (CsgIv*)lds + thread;
	st.tmp_ = (CsgIv*)lds + (size_t)cap * threads + thread;
	st.nodes_ = (float4*)((CsgIv*)lds + 2 * (size_t)cap * threads) + thread;
	st.threads = threads; st.cap_ = cap;
	return st;
}

// Both hits of a leaf at once (csg.cpp:99-111 asks rayhit for the entry and then for the exit): the two calls share the
// slab divisions (box) or the discriminant, its root and the quotients (sphere), so evaluating them together halves the
// work and gives the same two values.
__device__ __forceinline__ void csgLeafHits(const DevScene& S, const rt_csg_node& nd, f3 e, f3 d, float& tn, float& tf) {
	if (nd.kind == RT_CSG_SPHERE) {   // sphereHit(..., false) and sphereHit(..., true)
		const f3 ec = e - mk3(nd.f[0], nd.f[1], nd.f[2]);
		const float radius = nd.f[3];
		const float dde = dot(d, ec);
		const float dd = dot(d, d);
		const double disc = (double)dde * (double)dde - (double)(dd * (dot(ec, ec) - radius * radius));
		tn = tf = 0.0f;
		if (disc < 0) return;
		const double rest = (double)(dot(-d, ec) / dd);
		const double q = sqrt(disc) / (double)dd;
		const double t0 = rest - q, t1 = rest + q;
		if (!(t0 < 0)) tn = (float)t0;
		if (!(t1 < 0)) tf = (float)t1;
		return;
	}
	if (nd.kind == RT_CSG_BOX) {   // boxHit(..., false) and boxHit(..., true)
		float tnear = -RT_MAX_T, tfar = RT_MAX_T;
		bool miss = false;
		slab(nd.f[0], nd.f[1], e.x, d.x, tnear, tfar, miss);
		slab(nd.f[2], nd.f[3], e.y, d.y, tnear, tfar, miss);
		slab(nd.f[4], nd.f[5], e.z, d.z, tnear, tfar, miss);
		tn = (miss || tnear < 0) ? 0.0f : tnear;
		tf = (miss || tfar < 0) ? 0.0f : tfar;
		return;
	}
	tn = csgLeafHit(S, nd, e, d, false, nullptr);
	tf = csgLeafHit(S, nd, e, d, true, nullptr);
}

// std::sort of a list of more than 16 intervals (csg.cpp:199), kept out of line: no program of up to 16 sphere / box leaves gets here
template <class At>
__device__ __noinline__ void csgSortLong(At at, int n) {
	csgSortStd<CsgIv>(at, n, [](const CsgIv& x, const CsgIv& y) { return x.a.t < y.a.t; });
}

// Evaluates the object's post-order program for one ray; returns csgObject::rayhit's t (0 = no hit) and the tag of
// the selected boundary.  `cap` (intervals a list or the whole stack may hold) is either the full RTD_CSG_CAP or the
// scene's own bound, which no step of any of its programs can exceed: the capacity checks below then never fire and
// both stores give the same result.
template <class Store>
__device__ inline float csgEval(const DevScene& S, const rt_csg& c, f3 e, f3 d, bool exitHit, uint32_t& tagOut, Store& st) {
	const int cap = st.cap();
	int sp = 0;
	CsgStarts start;
	for (int k = 0; k < c.node_count; k++) {
		const rt_csg_node nd = st.node(S, c, k);
		if (nd.kind >= RT_CSG_SPHERE) {   // leaf, csg.cpp:99-111
			int at = start.get(sp);
			float tn, tf;
			csgLeafHits(S, nd, e, d, tn, tf);
			int n = 0;
			if (tf > 0 && at < cap && sp < RT_MAX_CSG_LEAVES) {
				CsgIv v;
				v.a.t = tn; v.a.tag = (uint32_t)k;
				v.b.t = tf; v.b.tag = (uint32_t)k | 0x100u;
				st.arena(at) = v;
				n = 1;
			}
			if (sp < RT_MAX_CSG_LEAVES) { sp++; start.set(sp, at + n); }
			continue;
		}
		if (sp < 2) continue;   // malformed program (rejected at upload); keeps the indices in range regardless
		const int a0 = start.get(sp - 2), a1 = start.get(sp - 1), b1 = start.get(sp);
		const int n1 = a1 - a0, n2 = b1 - a1;
		auto l1 = [&](int i) -> CsgIv& { return st.arena(a0 + i); };
		auto l2 = [&](int i) -> CsgIv& { return st.arena(a1 + i); };
		int n = 0;
		// The result replaces both operands at a0.  With an empty operand it is already there or empty; with a one-interval
		// first list (every step of a program whose right operands are the composite ones) that interval is held in a
		// register and the result written straight over the operands: output k lands on a slot of the second list that the
		// merge has passed (at most k + 1 - c1 of its elements are flushed when c2 are consumed) or on x's own.  Only longer
		// first lists go through `tmp`.  Same comparisons in the same order as the general form below, hence the same list.
		if (n1 == 0) n = nd.kind == RT_CSG_UNION ? n2 : 0;
		else if (n2 == 0 && nd.kind != RT_CSG_DIFFERENCE) n = nd.kind == RT_CSG_UNION ? n1 : 0;
		else if (n1 == 1) {
			const CsgIv x = l1(0);
			auto out = [&](int k) -> CsgIv& { return st.arena(a0 + k); };
			if (nd.kind == RT_CSG_UNION) {
				int c2 = 0;
				bool xLeft = true;
				CsgIv y = l2(0), cur;
				if (x.a.t < y.a.t) { cur = x; xLeft = false; } else { cur = y; c2 = 1; }
				while (xLeft || c2 < n2) {
					CsgIv add;
					if (c2 < n2) y = l2(c2);
					if (xLeft && c2 < n2) { if (x.a.t < y.a.t) { add = x; xLeft = false; } else { add = y; c2++; } }
					else if (xLeft) { add = x; xLeft = false; }
					else { add = y; c2++; }
					if (add.a.t < cur.b.t) cur.b = csgMaxB(cur.b, add.b);
					else { out(n++) = cur; cur = add; }
				}
				out(n++) = cur;
			} else if (nd.kind == RT_CSG_INTERSECTION) {
				for (int c2 = 0; c2 < n2; c2++) {
					const CsgIv y = l2(c2);
					if (x.a.t < y.b.t && x.b.t > y.a.t) {
						CsgIv v;
						v.a = csgMaxB(x.a, y.a);
						v.b = csgMinB(x.b, y.b);
						out(n++) = v;
					}
				}
				if (n > 16) csgSortLong(out, n);
				else
				for (int i = 1; i < n; i++) {
					const CsgIv v = out(i);
					int j = i;
					while (j > 0 && v.a.t < out(j - 1).a.t) { out(j) = out(j - 1); j--; }
					if (j != i) out(j) = v;
				}
			} else {
				CsgB cur = x.a;
				for (int c2 = 0; c2 < n2 && cur.t < x.b.t; c2++) {
					CsgIv sub = l2(c2);
					sub.a.tag ^= 0x200u; sub.b.tag ^= 0x200u;
					if (x.a.t < sub.b.t && x.b.t > sub.a.t) {
						if (!(sub.a.t < cur.t)) { CsgIv v; v.a = cur; v.b = sub.a; out(n++) = v; }
						cur = sub.b;
					}
				}
				if (cur.t < x.b.t) { CsgIv v; v.a = cur; v.b = x.b; out(n++) = v; }
			}
		} else {
			if (nd.kind == RT_CSG_UNION) {   // csg.cpp:123-188
				if (n1 == 0) { for (int i = 0; i < n2; i++) st.tmp(n++) = l2(i); }
				else if (n2 == 0) { for (int i = 0; i < n1; i++) st.tmp(n++) = l1(i); }
				else {
					int c1 = 0, c2 = 0;
					CsgIv cur;
					if (l1(0).a.t < l2(0).a.t) cur = l1(c1++); else cur = l2(c2++);
					while (c1 < n1 || c2 < n2) {
						CsgIv add;
						if (c1 < n1 && c2 < n2) { if (l1(c1).a.t < l2(c2).a.t) add = l1(c1++); else add = l2(c2++); }
						else if (c1 < n1) add = l1(c1++);
						else add = l2(c2++);
						if (add.a.t < cur.b.t) cur.b = csgMaxB(cur.b, add.b);
						else { if (n < cap) st.tmp(n++) = cur; cur = add; }
					}
					if (n < cap) st.tmp(n++) = cur;
				}
			} else if (nd.kind == RT_CSG_INTERSECTION) {   // csg.cpp:190-200
				for (int c1 = 0; c1 < n1; c1++) {
					const CsgIv x = l1(c1);
					for (int c2 = 0; c2 < n2; c2++) {
						const CsgIv y = l2(c2);
						if (x.a.t < y.b.t && x.b.t > y.a.t && n < cap) {
							CsgIv v;
							v.a = csgMaxB(x.a, y.a);
							v.b = csgMinB(x.b, y.b);
							st.tmp(n++) = v;
						}
					}
				}
				// std::sort by entry t; for <= 16 elements libstdc++ runs a plain (stable) insertion sort, beyond that its introsort
				// (csg_sort.h restates it: the order of equal entries is the library's)
				if (n > 16) csgSortLong([&](int i) -> CsgIv& { return st.tmp(i); }, n);
				else
				for (int i = 1; i < n; i++) {
					const CsgIv v = st.tmp(i);
					int j = i;
					while (j > 0 && v.a.t < st.tmp(j - 1).a.t) { st.tmp(j) = st.tmp(j - 1); j--; }
					st.tmp(j) = v;
				}
			} else {   // Difference, csg.cpp:202-230
				for (int c1 = 0; c1 < n1; c1++) {
					const CsgIv x = l1(c1);
					CsgB cur = x.a;
					for (int c2 = 0; c2 < n2 && cur.t < x.b.t; c2++) {
						CsgIv sub = l2(c2);
						sub.a.tag ^= 0x200u; sub.b.tag ^= 0x200u;
						if (x.a.t < sub.b.t && x.b.t > sub.a.t) {
							if (sub.a.t < cur.t) cur = sub.b;
							else {
								if (n < cap) { CsgIv v; v.a = cur; v.b = sub.a; st.tmp(n++) = v; }
								cur = sub.b;
							}
						}
					}
					if (cur.t < x.b.t && n < cap) { CsgIv v; v.a = cur; v.b = x.b; st.tmp(n++) = v; }
				}
			}
			if (a0 + n > cap) n = cap - a0;
			for (int i = 0; i < n; i++) st.arena(a0 + i) = st.tmp(i);
		}
		sp--;
		start.set(sp, a0 + n);
	}
	float t = 0;
	tagOut = 0;
	if (sp >= 1) {
		const int a0 = start.get(sp - 1), a1 = start.get(sp);
		for (int i = a0; i < a1 && t == 0; i++) {   // csg.cpp:13-22
			const CsgIv v = st.arena(i);
			if (!exitHit && v.a.t > 0) { t = v.a.t; tagOut = v.a.tag; }
			else if (exitHit && v.b.t > 0) { t = v.b.t; tagOut = v.b.tag; }
		}
	}
	return t;
}
__device__ inline float csgEval(const DevScene& S, const rt_csg& c, f3 e, f3 d, bool exitHit, uint32_t& tagOut) {
	CsgScratch st;
	return csgEval(S, c, e, d, exitHit, tagOut, st);
}

// The normal csgObject::getNormal would return (cachedHitNormal) for the boundary `tag` of the ray (e, d).
template <class Store>
__device__ inline f3 csgNormal(const DevScene& S, const rt_csg& c, f3 e, f3 d, uint32_t tag, const Store& st) {
	const rt_csg_node nd = st.node(S, c, (int)(tag & 0xffu));
	f3 n;
	csgLeafHit(S, nd, e, d, (tag & 0x100u) != 0, &n);
	return (tag & 0x200u) ? -n : n;
}
__device__ inline f3 csgNormal(const DevScene& S, const rt_csg& c, f3 e, f3 d, uint32_t tag) {
	CsgScratch st;
	return csgNormal(S, c, e, d, tag, st);
}

// EasyBMP BMP::operator()(i,j): both indices clamped (EasyBMP.cpp:310-329)
__device__ __forceinline__ const uint8_t* texel(const DevScene& S, int tex, int i, int j) {
	const rt_texture T = S.textures[tex];
	if (i >= T.width) i = T.width - 1;
	if (i < 0) i = 0;
	if (j >= T.height) j = T.height - 1;
	if (j < 0) j = 0;
	return S.texels + T.offset + 3 * ((int64_t)j * T.width + i);
}

// float -> int the way x86 cvttss2si does it for out-of-range / NaN inputs (INT_MIN), so that int(u*width) in
// bump.cpp:21 / texturemesh.cpp:12 agrees with the oracle even for garbage coordinates
__device__ __forceinline__ int f2i(float v) {
	if (!(v > -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000;
	return (int)v;
}

// Sphere::getNormal objects.cpp:180-182 / BumpSphere::getNormal bump.cpp:8-34
__device__ inline f3 sphereNormal(const DevScene& S, const rt_sphere& s, f3 hp) {
	f3 trueNormal = normalize(hp - ld3(s.center));
	if (s.bump_tex < 0) return trueNormal;
	// glibc's atan2f / asinf bit for bit (glibc_math.h): the texel indices below are truncations of u * width, v * height
	float u = (float)(0.5 - (double)rt_atan2f(-trueNormal.z, -trueNormal.x) / (2 * RTD_PI));
	float v = (float)(0.5 - (double)rt_asinf(trueNormal.y) / RTD_PI);
	const int height = S.textures[s.bump_tex].height, width = S.textures[s.bump_tex].width;
	float value = float(texel(S, s.bump_tex, f2i(u * width), f2i(v * height))[0]) / 255;
	float value_u = float(texel(S, s.bump_tex, f2i(u * width + 1) % width, f2i(v * height))[0]) / 255;
	float value_v = float(texel(S, s.bump_tex, f2i(u * width), f2i(v * height + 1) % height)[0]) / 255;
	float gradient_u = value_u - value;
	float gradient_v = value_v - value;
	f3 tangent_u = normalize(cross(mk3(0, 1, 0), trueNormal));
	f3 tangent_v = normalize(cross(trueNormal, tangent_u));
	return normalize(trueNormal + gradient_u * s.bump_depth * tangent_u + gradient_v * s.bump_depth * tangent_v);
}

// ---- raymath.cpp:3-24 ----
__device__ __forceinline__ bool refractRay(f3 Vi, f3 N, float refraction, f3& Vr) {
	float VidotN = dot(Vi, N);
	float refratio = (float)(1.0 / (double)refraction);
	f3 n = N;
	if (VidotN < 0) VidotN = -VidotN;
	else { refratio = (float)(1.0 / (double)refratio); n = -N; }
	float k = (float)(1 - ((double)refratio * (double)refratio) * (1 - (double)VidotN * (double)VidotN));
	if (k < 0) return false;
	Vr = normalize(refratio * Vi + (refratio * VidotN - sqrtf(k)) * n);
	return true;
}
__device__ __forceinline__ f3 reflectRay(f3 V, f3 N) { return normalize(2 * dot(N, V) * N - V); }

// ---- raymath.cpp:34-57: the two Phong terms for unit light colour I ----
__device__ __forceinline__ f3 phong(f3 I, const float* kd, const float* ks, float shininess, f3 N, f3 L, f3 V) {
	f3 colour = mk3(0, 0, 0);
	f3 diffuse = I * ld3(kd) * dot(N, L);
	if (diffuse.x < 0) diffuse.x = 0;
	if (diffuse.y < 0) diffuse.y = 0;
	if (diffuse.z < 0) diffuse.z = 0;
	colour = colour + diffuse;
	// A material without a specular colour (every wall of C4): I * ks is an exact zero per channel, and the power it would be multiplied
	// with is finite for an exponent in [0, 1e6] (its base is at most 1 + a few ulp), so the term adds zero: R, its normalisation and
	// powf are skipped.  A wave of the area-light kernel is one surface, so there the branch is uniform.
	const f3 Iks = I * ld3(ks);
	if (Iks.x == 0 && Iks.y == 0 && Iks.z == 0 && shininess >= 0.0f && shininess <= 1e6f) return colour;
	f3 R = normalize(2 * (dot(N, L)) * N - L);
	float RdotV = dot(R, V);
	if (RdotV > 0) {
		f3 specular = Iks * powf(RdotV, shininess);
		if (specular.x < 0) specular.x = 0;
		if (specular.y < 0) specular.y = 0;
		if (specular.z < 0) specular.z = 0;
		colour = colour + specular;
	}
	return colour;
}

}  // namespace rtd
