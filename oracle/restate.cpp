// oracle/restate.cpp -- TEST INFRASTRUCTURE ONLY: the CPU restatement of the reference's per-ray path.
//
// A deliberately literal, recursive, single-ray re-implementation of the reference's hot path over the flat
// scene IR (include/rt_scene.h), written to be compared bit-for-bit with the real reference
// (oracle/_ref/ref_driver) and then used as the checker for the HIP path.  It is NOT the product and the
// product never links it.  Parity status: PINNED by tests/golden/ref_*.npz (see restate.h).
//
// Build: g++ -O2 -ffp-contract=off, no -march=native / -ffast-math (an FMA changes results).
//
// Every function cites the reference lines it follows (paths relative to /root/reference/src).  Arithmetic
// keeps the reference's float/double mix and GLM 0.9.8.5's operation order (glm/detail/func_geometric.inl).
//
// Where the reference's result depends on hidden mutable state left by EARLIER rays (Object::cachedHitpoint
// after a failed exit search, objects.cpp:123-125), no stateless program can follow it; those cases are
// marked "STATE" below and use a fixed fallback.  They need a refracted ray that finds no exit on its own
// object (open meshes, extreme grazing) and are counted by the parity tests, not hidden.
//
// Area lights: with seed == 0 the samples come from libc rand() in the reference's call order
// (arealight.cpp:5,37-42,79-87), which reproduces oracle/_ref exactly when run single-threaded in ray order.
// With seed != 0 they come from the counter-based generator rng2() below, keyed by (seed, ray index, position
// in the ray tree, light, sample); the device uses the same function, so device and oracle see the same
// sample points.
#include "restate.h"

#include <atomic>
#include <unordered_map>

// exit searches of transmitted rays that found no exit (the "STATE" cases above), since the last reset
static std::atomic<uint64_t> g_stateCases{0};
// Textured triangles lit with another hit's texel (see trace() below), since the last reset; and whether that behaviour of the
// reference is reproduced (default) or every hit is lit with its own texel (what the device's wavefront pipeline does).
static std::atomic<uint64_t> g_texelCases{0};
static std::atomic<int> g_texelState{1};

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---- GLM-ordered float3 ------------------------------------------------------------------------------
struct V3 {
	float x, y, z;
	V3() : x(0), y(0), z(0) {}
	V3(float a, float b, float c) : x(a), y(b), z(c) {}
	explicit V3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
	float operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline float dot(V3 a, V3 b) { V3 t = a * b; return t.x + t.y + t.z; }                      // func_geometric.inl:54-60
inline V3 cross(V3 x, V3 y) { return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }  // :74-85
inline float length(V3 a) { return std::sqrt(dot(a, a)); }                                    // :14-19
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }                      // :88-96 (inversesqrt = 1/sqrt)
inline bool isZero(const float* v) { return v[0] == 0 && v[1] == 0 && v[2] == 0; }            // raymath.cpp:59-61

const double PI = 3.14159265358979323846264338327950288;   // objects.h:9

// ---- counter-based replacement for rand() -------------------------------------------------------------
inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
inline void rng2(uint32_t seed, uint32_t ray, uint32_t path, uint32_t light, uint32_t sample, float& u1, float& u2) {
	uint32_t h = mix32(seed ^ 0x9e3779b9u);
	h = mix32(h ^ ray);
	h = mix32(h + path * 0x85ebca6bu);
	h = mix32(h ^ (light * 0xc2b2ae35u + sample));
	uint32_t a = mix32(h ^ 0x27d4eb2fu), b = mix32(h + 0x165667b1u);
	u1 = (float)(a >> 8) * (1.0f / 16777216.0f);
	u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
}

struct Hit {
	uint32_t prim = RT_NO_HIT;   // RT_PRIM_REF
	float t = 0;
	V3 point;                    // the winner's cachedHitpoint
	V3 normal;                   // csgObject::cachedHitNormal (csg hits only)
};

struct Tracer {
	const rt_flat_scene* S;
	int maxDepth;
	uint32_t seed;
	uint32_t rayIndex = 0;
	uint64_t nNear = 0, nShadow = 0;

	// ---- BoundingBox::intersect, objects.cpp:11-44 ----
	static float boxIntersect(const float* b, V3 e, V3 d, bool exit) {
		float tnear = -RT_MAX_T, tfar = RT_MAX_T;
		float mn[3] = { b[0], b[2], b[4] }, mx[3] = { b[1], b[3], b[5] };
		for (int axis = 0; axis < 3; axis++) {
			if (d[axis] == 0 && (e[axis] < mn[axis] || e[axis] > mx[axis])) return -1;
			float t1 = (mn[axis] - e[axis]) / d[axis];
			float t2 = (mx[axis] - e[axis]) / d[axis];
			if (t1 > t2) std::swap(t1, t2);
			if (t1 > tnear) tnear = t1;
			if (t2 < tfar) tfar = t2;
			if (tnear > tfar) return -1;
			if (tfar < 0) return -1;
		}
		if (!exit) return tnear < 0 ? 0 : tnear;
		return tfar < 0 ? 0 : tfar;
	}

	// ---- Sphere::rayhit, objects.cpp:159-178.  `hp` is cachedHitpoint: written only on success. ----
	static float sphereHit(V3 center, float radius, V3 e, V3 d, bool exit, V3& hp) {
		double disc = std::pow((double)dot(d, e - center), 2) - (double)(dot(d, d) * (dot(e - center, e - center) - radius * radius));
		if (disc < 0) return 0;
		double rest = dot(-d, e - center) / dot(d, d);
		double t;
		if (!exit) t = rest - std::sqrt(disc) / dot(d, d);
		else t = rest + std::sqrt(disc) / dot(d, d);
		if (t < 0) return 0;
		hp = e + float(t) * d;
		return float(t);
	}

	// ---- Plane::rayhit, objects.cpp:199-213 ----
	static float planeHit(V3 point, V3 normal, V3 e, V3 d, bool exit, V3& hp) {
		V3 n = normal;
		if (exit) n = -normal;
		double numerator = dot(n, point - e);
		double denominator = dot(n, d);
		double t = numerator / denominator;
		if (t <= 0 || numerator > 0) return 0;
		hp = e + float(t) * d;
		return float(t);
	}

	// ---- pointInTriangle, raymath.cpp:26-32 ----
	static bool pointInTriangle(V3 point, V3 p1, V3 p2, V3 p3, V3 n) {
		float test1 = dot(cross(point - p1, p2 - p1), n);
		float test2 = dot(cross(point - p2, p3 - p2), n);
		float test3 = dot(cross(point - p3, p1 - p3), n);
		return (test1 >= 0 && test2 >= 0 && test3 >= 0) || (test1 <= 0 && test2 <= 0 && test3 <= 0);
	}

	// ---- Triangle::rayhit, objects.cpp:251-261 ----
	static float triHit(const rt_tri& tr, V3 e, V3 d, bool exit, V3& hp) {
		V3 scratch;
		float t = planeHit(V3(tr.p0), V3(tr.n), e, d, exit, scratch);
		V3 hitpos = e + t * d;
		if (t <= 0 || !pointInTriangle(hitpos, V3(tr.p0), V3(tr.p1), V3(tr.p2), V3(tr.n))) return 0;
		hp = e + t * d;
		return t;
	}

	// ---- Mesh::rayhit, objects.cpp:286-304.  hp / nrm are Mesh::cachedHitpoint / cachedHitNormal. ----
	float meshHit(const rt_mesh& m, V3 e, V3 d, bool exit, V3& hp, V3& nrm) const {
		float t_min = RT_MAX_T;
		for (int i = 0; i < m.tri_count; i++) {
			const rt_tri& tr = S->mesh_tris[m.tri_first + i];
			V3 scratch;
			float t = triHit(tr, e, d, exit, scratch);
			V3 hitpos = e + t * d;
			if (t > 0.0 && t < t_min) { hp = hitpos; nrm = V3(tr.n); t_min = t; }
		}
		if (t_min == RT_MAX_T) return 0.0;
		return t_min;
	}

	// ---- Box::rayhit / Box::getNormal, objects.cpp:337-361 ----
	static float boxHit(const float* b, V3 e, V3 d, bool exit, V3& hp) {
		float t = boxIntersect(b, e, d, exit);
		if (t < 0) return 0;
		hp = e + t * d;
		return t;
	}
	static void boxNormal(const float* b, V3 hp, V3& n) {   // leaves n untouched when no face matches
		if (std::fabs(hp.x - b[0]) < 1e-5) n = V3(-1, 0, 0);
		else if (std::fabs(hp.x - b[1]) < 1e-5) n = V3(1, 0, 0);
		else if (std::fabs(hp.y - b[2]) < 1e-5) n = V3(0, -1, 0);
		else if (std::fabs(hp.y - b[3]) < 1e-5) n = V3(0, 1, 0);
		else if (std::fabs(hp.z - b[4]) < 1e-5) n = V3(0, 0, -1);
		else if (std::fabs(hp.z - b[5]) < 1e-5) n = V3(0, 0, 1);
	}

	// ---- CSG: csg.h:8-13, csg_node::setIntervals csg.cpp:95-232, csgObject::rayhit csg.cpp:8-26 ----
	struct Isect { float t = 0; V3 normal; };
	struct Interval { Isect a[2]; Isect& operator[](int i) { return a[i]; } const Isect& operator[](int i) const { return a[i]; } };
	static bool compareIntervals(const Interval& i1, const Interval& i2) { return i1[0].t < i2[0].t; }
	static bool compareIntersections(const Isect& i1, const Isect& i2) { return i1.t < i2.t; }

	void csgLeaf(const rt_csg_node& nd, V3 e, V3 d, std::vector<Interval>& out) const {   // csg.cpp:99-111
		Isect nearI, farI;
		V3 hp;   // the leaf object's cachedHitpoint.  When rayhit fails it keeps an older value in the reference, but a
		         // failed rayhit returns t = 0 and a t = 0 boundary can never be selected (csg.cpp:14,18), so its normal is moot.
		if (nd.kind == RT_CSG_SPHERE) {
			V3 c(nd.f[0], nd.f[1], nd.f[2]);
			nearI.t = sphereHit(c, nd.f[3], e, d, false, hp); nearI.normal = normalize(hp - c);   // Sphere::getNormal objects.cpp:180-182
			farI.t = sphereHit(c, nd.f[3], e, d, true, hp); farI.normal = normalize(hp - c);
		} else if (nd.kind == RT_CSG_BOX) {
			nearI.t = boxHit(nd.f, e, d, false, hp); boxNormal(nd.f, hp, nearI.normal);
			farI.t = boxHit(nd.f, e, d, true, hp); boxNormal(nd.f, hp, farI.normal);
		} else {
			const rt_mesh& m = S->meshes[nd.mesh];
			V3 nrm;
			nearI.t = meshHit(m, e, d, false, hp, nrm); nearI.normal = nrm;
			farI.t = meshHit(m, e, d, true, hp, nrm); farI.normal = nrm;
		}
		if (farI.t > 0) { Interval iv; iv[0] = nearI; iv[1] = farI; out.push_back(iv); }
	}

	static void csgCombine(int op, const std::vector<Interval>& list1, const std::vector<Interval>& list2, std::vector<Interval>& intervalList) {
		size_t count1 = 0, count2 = 0;
		if (op == RT_CSG_UNION) {   // csg.cpp:123-188
			if (list1.size() == 0) intervalList = list2;
			else if (list2.size() == 0) intervalList = list1;
			else {
				Interval current;
				if (compareIntervals(list1[0], list2[0])) { current = list1[0]; count1++; }
				else { current = list2[0]; count2++; }
				while (count1 < list1.size() && count2 < list2.size()) {
					Interval add;
					if (compareIntervals(list1[count1], list2[count2])) { add = list1[count1]; count1++; }
					else { add = list2[count2]; count2++; }
					if (add[0].t < current[1].t) current[1] = std::max(current[1], add[1], compareIntersections);
					else { intervalList.push_back(current); current = add; }
				}
				while (count1 < list1.size()) {
					if (list1[count1][0].t < current[1].t) current[1] = std::max(current[1], list1[count1][1], compareIntersections);
					else { intervalList.push_back(current); current = list1[count1]; }
					count1++;
				}
				while (count2 < list2.size()) {
					if (list2[count2][0].t < current[1].t) current[1] = std::max(current[1], list2[count2][1], compareIntersections);
					else { intervalList.push_back(current); current = list2[count2]; }
					count2++;
				}
				intervalList.push_back(current);
			}
		}
		if (op == RT_CSG_INTERSECTION) {   // csg.cpp:190-200
			for (count1 = 0; count1 < list1.size(); count1++)
				for (count2 = 0; count2 < list2.size(); count2++)
					if (list1[count1][0].t < list2[count2][1].t && list1[count1][1].t > list2[count2][0].t) {
						Interval iv;
						iv[0] = std::max(list1[count1][0], list2[count2][0], compareIntersections);
						iv[1] = std::min(list1[count1][1], list2[count2][1], compareIntersections);
						intervalList.push_back(iv);
					}
			std::sort(intervalList.begin(), intervalList.end(), compareIntervals);
		}
		if (op == RT_CSG_DIFFERENCE) {   // csg.cpp:202-230
			Isect current;
			for (count1 = 0; count1 < list1.size(); count1++) {
				current = list1[count1][0];
				for (count2 = 0; count2 < list2.size() && current.t < list1[count1][1].t; count2++) {
					Interval subtract = list2[count2];
					subtract[0].normal = -subtract[0].normal;
					subtract[1].normal = -subtract[1].normal;
					if (list1[count1][0].t < subtract[1].t && list1[count1][1].t > subtract[0].t) {
						if (subtract[0].t < current.t) current = subtract[1];
						else {
							Interval iv; iv[0] = current; iv[1] = subtract[0];
							intervalList.push_back(iv);
							current = subtract[1];
						}
					}
				}
				if (current.t < list1[count1][1].t) { Interval iv; iv[0] = current; iv[1] = list1[count1][1]; intervalList.push_back(iv); }
			}
		}
	}

	// csgObject::rayhit, csg.cpp:8-26.  hp = cachedHitpoint (always rewritten), nrm = cachedHitNormal (only on a hit).
	float csgHit(const rt_csg& c, V3 e, V3 d, bool exit, V3& hp, V3& nrm) const {
		std::vector<std::vector<Interval>> stack;
		for (int k = 0; k < c.node_count; k++) {
			const rt_csg_node& nd = S->csg_nodes[c.node_first + k];
			if (nd.kind >= RT_CSG_SPHERE) { stack.emplace_back(); csgLeaf(nd, e, d, stack.back()); }
			else {
				std::vector<Interval> second = std::move(stack.back()); stack.pop_back();
				std::vector<Interval> first = std::move(stack.back()); stack.pop_back();
				stack.emplace_back();
				csgCombine(nd.kind, first, second, stack.back());
			}
		}
		const std::vector<Interval>& root = stack.back();
		float t = 0;
		for (size_t i = 0; i < root.size() && t == 0; i++) {
			if (!exit && root[i][0].t > 0) { t = root[i][0].t; nrm = root[i][0].normal; }
			else if (exit && root[i][1].t > 0) { t = root[i][1].t; nrm = root[i][1].normal; }
		}
		hp = e + t * d;
		return t;
	}

	// object->rayhit(e, d) for a BVH primitive; fills the candidate hit record
	float primHit(uint32_t ref, V3 e, V3 d, V3& hp, V3& nrm) const {
		uint32_t ix = RT_PRIM_INDEX(ref);
		switch (RT_PRIM_TYPE(ref)) {
		case RT_PRIM_SPHERE: { const rt_sphere& s = S->spheres[ix]; return sphereHit(V3(s.center), s.radius, e, d, false, hp); }
		case RT_PRIM_TRI: return triHit(S->tris[ix], e, d, false, hp);
		case RT_PRIM_BOX: return boxHit(S->boxes[ix].b, e, d, false, hp);
		case RT_PRIM_CSG: return csgHit(S->csgs[ix], e, d, false, hp, nrm);
		}
		return 0;
	}

	// ---- BVH::findNearest / findRecursive, bvh.cpp:72-121 ----
	float findRecursive(const float* box, int32_t child, int32_t count, V3 e, V3 d, float t_min, Hit& hit) const {
		float t = boxIntersect(box, e, d, false);
		if (t < 0 || t > t_min) return t_min;
		if (child >= 0) {
			const rt_bvh_node& n = S->bvh_nodes[child];
			t_min = findRecursive(n.box[0], n.child[0], n.count[0], e, d, t_min, hit);
			t_min = findRecursive(n.box[1], n.child[1], n.count[1], e, d, t_min, hit);
		} else {
			int32_t first = ~child;
			for (int i = 0; i < count; i++) {
				uint32_t ref = S->leaf_prims[first + i] & ~RT_PRIM_LAST;
				V3 hp, nrm;
				t = primHit(ref, e, d, hp, nrm);
				if (t > 1e-5 && t < t_min) { hit.prim = ref; hit.t = t; hit.point = hp; hit.normal = nrm; t_min = t; }
			}
		}
		return t_min;
	}
	Hit findNearest(V3 e, V3 d) {
		nNear++;
		float t_min = RT_MAX_T;
		Hit hit;
		for (int i = 0; i < S->n_planes; i++) {
			const rt_plane& p = S->planes[i];
			V3 hp;
			float t = planeHit(V3(p.point), V3(p.normal), e, d, false, hp);
			if (t > 0 && t < t_min) { hit.prim = RT_PRIM_REF(RT_PRIM_PLANE, i); hit.t = t; hit.point = hp; t_min = t; }
		}
		if (!(S->root_child < 0 && S->root_count == 0))
			findRecursive(S->root_box, S->root_child, S->root_count, e, d, t_min, hit);
		return hit;
	}

	// ---- BVH::calcShadow / shadowRecursive, bvh.cpp:123-160 ----
	bool shadowRecursive(const float* box, int32_t child, int32_t count, V3 e, V3 d, V3& shadow) const {
		float t = boxIntersect(box, e, d, false);
		if (t < 0 || t > 1) return true;
		if (child >= 0) {
			const rt_bvh_node& n = S->bvh_nodes[child];
			if (shadowRecursive(n.box[0], n.child[0], n.count[0], e, d, shadow) == false) return false;
			if (shadowRecursive(n.box[1], n.child[1], n.count[1], e, d, shadow) == false) return false;
		} else {
			int32_t first = ~child;
			for (int i = 0; i < count; i++) {
				uint32_t ref = S->leaf_prims[first + i] & ~RT_PRIM_LAST;
				V3 hp, nrm;
				float t = primHit(ref, e, d, hp, nrm);
				if (t < 1.0 && t * length(d) > 1e-5) {
					const rt_material& m = S->materials[primMaterial(ref)];
					if (!isZero(m.transmissive)) shadow = shadow * V3(m.transmissive);
					else return false;
				}
			}
		}
		return true;
	}
	bool calcShadow(V3 point, V3 lightPos, V3& shadow) {
		nShadow++;
		V3 direction = lightPos - point;
		if (S->root_child < 0 && S->root_count == 0) return true;
		return shadowRecursive(S->root_box, S->root_child, S->root_count, point, direction, shadow);
	}

	int primMaterial(uint32_t ref) const {
		uint32_t ix = RT_PRIM_INDEX(ref);
		switch (RT_PRIM_TYPE(ref)) {
		case RT_PRIM_SPHERE: return S->spheres[ix].mat;
		case RT_PRIM_TRI: return S->tris[ix].mat;
		case RT_PRIM_BOX: return S->boxes[ix].mat;
		case RT_PRIM_CSG: return S->csgs[ix].mat;
		case RT_PRIM_PLANE: return S->planes[ix].mat;
		}
		return 0;
	}
	int primId(uint32_t ref) const {
		uint32_t ix = RT_PRIM_INDEX(ref);
		switch (RT_PRIM_TYPE(ref)) {
		case RT_PRIM_SPHERE: return S->spheres[ix].id;
		case RT_PRIM_TRI: return S->tris[ix].id;
		case RT_PRIM_BOX: return S->boxes[ix].id;
		case RT_PRIM_CSG: return S->csgs[ix].id;
		case RT_PRIM_PLANE: return S->planes[ix].id;
		}
		return -1;
	}

	// EasyBMP BMP::operator()(i,j): clamps both indices (EasyBMP.cpp:310-329)
	const uint8_t* texel(int tex, int i, int j) const {
		const rt_texture& T = S->textures[tex];
		if (i >= T.width) i = T.width - 1;
		if (i < 0) i = 0;
		if (j >= T.height) j = T.height - 1;
		if (j < 0) j = 0;
		return S->texels + T.offset + 3 * ((int64_t)j * T.width + i);
	}

	// ---- raymath.cpp:3-24 ----
	static bool refractRay(V3 Vi, V3 N, float refraction, V3& Vr) {
		float VidotN = dot(Vi, N);
		float refratio = 1.0 / refraction;
		V3 n = N;
		if (VidotN < 0) VidotN = -VidotN;
		else { refratio = 1.0 / refratio; n = -N; }
		float k = 1 - std::pow((double)refratio, 2) * (1 - std::pow((double)VidotN, 2));
		if (k < 0) return false;
		Vr = normalize(refratio * Vi + (refratio * VidotN - std::sqrt(k)) * n);
		return true;
	}
	static void reflectRay(V3 V, V3 N, V3& R) { R = normalize(2 * dot(N, V) * N - V); }

	// ---- raymath.cpp:34-57 ----
	static void addDiffuse(V3 Id, V3 Kd, V3 N, V3 L, V3& colour) {
		V3 diffuse = Id * Kd * dot(N, L);
		if (diffuse.x < 0) diffuse.x = 0;
		if (diffuse.y < 0) diffuse.y = 0;
		if (diffuse.z < 0) diffuse.z = 0;
		colour = colour + diffuse;
	}
	static void addSpecular(V3 Is, V3 Ks, float a, V3 N, V3 L, V3 V, V3& colour) {
		V3 R = normalize(2 * (dot(N, L)) * N - L);
		float RdotV = dot(R, V);
		if (RdotV > 0) {
			V3 specular = Is * Ks * std::pow(RdotV, a);   // float pow, as on MSVC / with `using std::pow`
			if (specular.x < 0) specular.x = 0;
			if (specular.y < 0) specular.y = 0;
			if (specular.z < 0) specular.z = 0;
			colour = colour + specular;
		}
	}

	// ---- lights: objects.cpp:382-458, arealight.cpp:7-42,79-87 ----
	void lightPoint(const rt_light& L, int lightIndex, uint32_t path, V3 p, V3 N, V3 V, const rt_material& m, V3& pointColour) {
		V3 colour(L.colour);
		switch (L.type) {
		case RT_LIGHT_AMBIENT: pointColour = pointColour + colour * V3(m.ambient); break;
		case RT_LIGHT_DIRECTIONAL: {
			V3 Ld = -V3(L.direction);
			V3 lightPosition = p + float(RT_MAX_T) * Ld;
			V3 shadow(1.0, 1.0, 1.0);
			if (calcShadow(p, lightPosition, shadow)) {
				V3 I = colour * shadow;
				addDiffuse(I, V3(m.diffuse), N, Ld, pointColour);
				addSpecular(I, V3(m.specular), m.shininess, N, Ld, V, pointColour);
			}
			break; }
		case RT_LIGHT_POINT: {
			V3 shadow(1.0, 1.0, 1.0), position(L.position);
			if (calcShadow(p, position, shadow)) {
				V3 I = colour * shadow;
				V3 Ld = normalize(position - p);
				addDiffuse(I, V3(m.diffuse), N, Ld, pointColour);
				addSpecular(I, V3(m.specular), m.shininess, N, Ld, V, pointColour);
			}
			break; }
		case RT_LIGHT_SPOT: {
			V3 shadow(1.0, 1.0, 1.0), position(L.position);
			if (calcShadow(p, position, shadow)) {
				V3 dir = -V3(L.direction);
				V3 Ld = normalize(position - p);
				if (dot(Ld, dir) > L.cos_cutoff) {
					V3 I = colour * shadow;
					addDiffuse(I, V3(m.diffuse), N, Ld, pointColour);
					addSpecular(I, V3(m.specular), m.shininess, N, Ld, V, pointColour);
				}
			}
			break; }
		case RT_LIGHT_RECT: case RT_LIGHT_CIRC: {
			V3 totalColour(0.0, 0.0, 0.0);
			for (int i = 0; i < L.samples; i++) {
				float r1, r2;
				if (seed == 0) { r1 = (float)rand() / RAND_MAX; r2 = (float)rand() / RAND_MAX; }
				else rng2(seed, rayIndex, path, (uint32_t)lightIndex, (uint32_t)i, r1, r2);
				V3 lightPos;
				if (L.type == RT_LIGHT_RECT) {
					float xDisplacement = (r1 - 0.5) * L.width;
					float yDisplacement = (r2 - 0.5) * L.height;
					lightPos = V3(L.position) + V3(L.planeX) * xDisplacement + V3(L.planeY) * yDisplacement;
				} else {
					float angle = r1 * 2 * PI;
					float displacement = L.radius * std::sqrt(r2);
					float xDisplacement = displacement * std::cos(angle);
					float yDisplacement = displacement * std::sin(angle);
					lightPos = V3(L.position) + V3(L.planeX) * xDisplacement + V3(L.planeY) * yDisplacement;
				}
				V3 shadow(1.0, 1.0, 1.0);
				if (calcShadow(p, lightPos, shadow)) {
					V3 I = colour * shadow;
					V3 Ld = normalize(lightPos - p);
					addDiffuse(I, V3(m.diffuse), N, Ld, totalColour);
					addSpecular(I, V3(m.specular), m.shininess, N, Ld, V, totalColour);
				}
			}
			pointColour = pointColour + totalColour / float(L.samples);
			break; }
		}
	}

	// ---- getNormal of the hit object ----
	V3 sphereNormal(const rt_sphere& s, V3 hp) const {
		V3 trueNormal = normalize(hp - V3(s.center));   // Sphere::getNormal, objects.cpp:180-182
		if (s.bump_tex < 0) return trueNormal;
		// BumpSphere::getNormal, bump.cpp:8-34
		float u = 0.5 - std::atan2(-trueNormal.z, -trueNormal.x) / (2 * PI);
		float v = 0.5 - std::asin(trueNormal.y) / PI;
		int height = S->textures[s.bump_tex].height, width = S->textures[s.bump_tex].width;
		float value = float(texel(s.bump_tex, int(u * width), int(v * height))[0]) / 255;
		float value_u = float(texel(s.bump_tex, int(u * width + 1) % width, int(v * height))[0]) / 255;
		float value_v = float(texel(s.bump_tex, int(u * width), int(v * height + 1) % height)[0]) / 255;
		float gradient_u = value_u - value;
		float gradient_v = value_v - value;
		V3 tangent_u = normalize(cross(V3(0, 1, 0), trueNormal));
		V3 tangent_v = normalize(cross(trueNormal, tangent_u));
		return normalize(trueNormal + gradient_u * s.bump_depth * tangent_u + gradient_v * s.bump_depth * tangent_v);
	}

	// ---- Object::transmitRay objects.cpp:107-140, Plane::transmitRay :224-229, Triangle::transmitRay :273-275 ----
	bool transmitRay(uint32_t ref, const rt_material& m, V3 inPoint, V3 inVector, V3 inNormal, V3& outPoint, V3& outVector) const {
		uint32_t type = RT_PRIM_TYPE(ref), ix = RT_PRIM_INDEX(ref);
		if (type == RT_PRIM_PLANE || m.refraction == 0) {
			outVector = inVector;
			outPoint = inPoint + 1e-5f * outVector;
			return true;
		}
		V3 currentPoint = inPoint, innerVector, outNormal;
		// csgObject::getNormal returns the member cachedHitNormal, which a failed exit search does not touch: it still holds the normal
		// of the last successful rayhit on the object -- the hit this ray entered through, unless a shadow ray of the same hit went
		// through the object in between (STATE).  (The other classes write into the caller's uninitialised local.)
		if (type == RT_PRIM_CSG) outNormal = inNormal;
		refractRay(inVector, inNormal, m.refraction, innerVector);
		bool result = false;
		int reflections = 0;
		while (!result && reflections < RT_MAX_REFLECTIONS) {
			// rayhit(currentPoint, innerVector, true); outPoint = cachedHitpoint; getNormal(outNormal)
			V3 hp = currentPoint;   // STATE: a failed exit search leaves an older cachedHitpoint in the reference; fixed fallback here
			float tExit = 0;
			switch (type) {
			case RT_PRIM_SPHERE: { const rt_sphere& s = S->spheres[ix]; tExit = sphereHit(V3(s.center), s.radius, currentPoint, innerVector, true, hp); outNormal = sphereNormal(s, hp); break; }
			case RT_PRIM_TRI: {
				// Mesh::cachedHitpoint / cachedHitNormal of a top-level mesh are written by exit searches only, and a failed one goes on
				// from what the last successful one left -- of an EARLIER ray, possibly an earlier pixel (STATE).  With seed 0 (one thread,
				// rays in the reference's order: the mode that is compared with the reference) that state is carried along; otherwise, and
				// on the device, the fallback is the current point and the previous normal.
				const int mi = S->tris[ix].mesh;
				V3 nrm = outNormal;
				if (seed == 0) { auto it = meshCache.find(mi); if (it != meshCache.end()) { hp = it->second.first; nrm = it->second.second; } }
				tExit = meshHit(S->meshes[mi], currentPoint, innerVector, true, hp, nrm);
				if (seed == 0 && tExit > 0) meshCache[mi] = std::make_pair(hp, nrm);
				outNormal = nrm; break; }
			case RT_PRIM_BOX: tExit = boxHit(S->boxes[ix].b, currentPoint, innerVector, true, hp); boxNormal(S->boxes[ix].b, hp, outNormal); break;
			case RT_PRIM_CSG: { V3 nrm = outNormal; tExit = csgHit(S->csgs[ix], currentPoint, innerVector, true, hp, nrm); outNormal = nrm; break; }   // STATE on a miss
			}
			if (!(tExit > 0)) g_stateCases.fetch_add(1, std::memory_order_relaxed);   // counted, not hidden: oracle_state_cases()
			outPoint = hp;
			result = refractRay(innerVector, outNormal, m.refraction, outVector);
			if (!result) {
				V3 R;
				reflectRay(-innerVector, outNormal, R);
				innerVector = R;
				currentPoint = outPoint;
				reflections++;
			}
		}
		return result;
	}

	mutable std::unordered_map<int, std::pair<V3, V3>> meshCache;   // Mesh::cachedHitpoint / cachedHitNormal (seed 0 only, see transmitRay)
	std::unordered_map<uint32_t, V3> texelOf;   // TextureTriangle::material.ambient / .diffuse as the last lightPoint on it left them

	// ---- trace raytracer.cpp:307-329 + Object::lightPoint objects.cpp:54-105 + TextureTriangle::lightPoint texturemesh.cpp:25-52 ----
	bool trace(V3 e, V3 s, V3& colour, int reflectionCount, uint32_t path, Hit* firstHit) {
		if (reflectionCount > maxDepth) { colour = V3(0, 0, 0); return false; }
		V3 d = s - e;
		Hit hit = findNearest(e, d);
		if (firstHit) *firstHit = hit;
		if (hit.prim == RT_NO_HIT) return false;

		uint32_t type = RT_PRIM_TYPE(hit.prim), ix = RT_PRIM_INDEX(hit.prim);
		rt_material material = S->materials[primMaterial(hit.prim)];
		V3 p = hit.point;
		bool textured = false;

		if (type == RT_PRIM_TRI && S->tris[ix].uv >= 0) {   // TextureTriangle::lightPoint
			const rt_tri& tr = S->tris[ix];
			const rt_uv& uvs = S->uvs[tr.uv];
			V3 P0(tr.p0), P1(tr.p1), P2(tr.p2);
			V3 v0 = P0 - p, v1 = P1 - p, v2 = P2 - p;
			float area = length(cross(P0 - P1, P0 - P2));
			float a0 = length(cross(v1, v2)) / area;
			float a1 = length(cross(v2, v0)) / area;
			float a2 = length(cross(v0, v1)) / area;
			float u = (uvs.uv[0] * a0 + uvs.uv[2] * a1) + uvs.uv[4] * a2;
			float v = (uvs.uv[1] * a0 + uvs.uv[3] * a1) + uvs.uv[5] * a2;
			int tex = S->meshes[tr.mesh].tex;
			int height = S->textures[tex].height, width = S->textures[tex].width;
			const uint8_t* px = texel(tex, int(u * width), int(v * height));   // TextureMesh::getTexValue, texturemesh.cpp:7-16
			V3 texColour(float(px[0]) / 255, float(px[1]) / 255, float(px[2]) / 255);
			material.ambient[0] = texColour.x; material.ambient[1] = texColour.y; material.ambient[2] = texColour.z;
			material.diffuse[0] = texColour.x; material.diffuse[1] = texColour.y; material.diffuse[2] = texColour.z;
			// texturemesh.cpp:49-50 writes the TRIANGLE's own material: state that outlives this call.  It is only ever read back
			// by a hit on the same triangle whose reflected ray has been traced in between, i.e. when the material reflects.
			if (!isZero(material.reflective)) { texelOf[ix] = texColour; textured = true; }
		}

		V3 V = normalize(-d);
		V3 N;
		switch (type) {
		case RT_PRIM_SPHERE: N = sphereNormal(S->spheres[ix], p); break;
		case RT_PRIM_PLANE: N = normalize(V3(S->planes[ix].normal)); break;   // objects.cpp:215-217
		case RT_PRIM_TRI: N = V3(S->tris[ix].n); break;
		case RT_PRIM_BOX: boxNormal(S->boxes[ix].b, p, N); break;
		case RT_PRIM_CSG: N = hit.normal; break;
		}

		colour = V3(0.0, 0.0, 0.0);
		if (!isZero(material.reflective)) {
			V3 R;
			reflectRay(V, N, R);
			bool h = trace(p + float(1e-5) * R, p + R, colour, reflectionCount + 1, path * 2, nullptr);
			if (!h) colour = V3(S->background);
			colour = colour * V3(material.reflective);
			// The reflected ray's subtree may have come back to this very triangle (two reflective meshes facing each other): its
			// lightPoint overwrote material.ambient / .diffuse (texturemesh.cpp:49-50), and the lights below (objects.cpp:77-79) read
			// the member, i.e. the texel of the LAST hit on the triangle -- state of this ray tree only, since every hit sets it first.
			if (textured && g_texelState.load(std::memory_order_relaxed)) {
				const V3 cur = texelOf[ix];
				if (cur.x != material.diffuse[0] || cur.y != material.diffuse[1] || cur.z != material.diffuse[2]) {
					g_texelCases.fetch_add(1, std::memory_order_relaxed);
					material.ambient[0] = cur.x; material.ambient[1] = cur.y; material.ambient[2] = cur.z;
					material.diffuse[0] = cur.x; material.diffuse[1] = cur.y; material.diffuse[2] = cur.z;
				}
			}
		}
		for (int i = 0; i < S->n_lights; i++) lightPoint(S->lights[i], i, path, p, N, V, material, colour);
		if (!isZero(material.transmissive)) {
			V3 transcolour(0.0, 0.0, 0.0), transmissionOrigin, transmissionDirection;
			bool result = transmitRay(hit.prim, material, p, d, N, transmissionOrigin, transmissionDirection);
			if (result) {
				bool h = trace(transmissionOrigin, transmissionOrigin + transmissionDirection, transcolour, reflectionCount + 1, path * 2 + 1, nullptr);
				if (!h) transcolour = V3(S->background);
			}
			colour = (V3(1.0, 1.0, 1.0) - V3(material.transmissive)) * colour + V3(material.transmissive) * transcolour;
		}
		return true;
	}
};

V3 primaryTarget(const rt_camera* c, int W, int H, int x, int y, int aaX, int aaY, int k) {   // q1.cpp:51-64
	float ox = ((k % aaX) + 0.5f) / aaX, oy = ((k / aaX) + 0.5f) / aaY;
	return V3(c->eye) + V3(c->facing) + V3(c->right) * (2 * ((x + ox) / W - 0.5f)) + V3(c->up) * (2 * ((y + oy) / H - 0.5f));
}

template <class F> void parallelFor(size_t n, int threads, F f) {
	if (threads <= 1 || n < 2) { f(0, n, 0); return; }
	std::vector<std::thread> pool;
	size_t chunk = (n + threads - 1) / threads;
	for (int t = 0; t < threads; t++) {
		size_t a = std::min(n, (size_t)t * chunk), b = std::min(n, a + chunk);
		if (a < b) pool.emplace_back([=] { f(a, b, t); });
	}
	for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

void oracle_primary_target(const rt_camera* camera, int width, int height, int x, int y, int aa_x, int aa_y, int k, float out[3]) {
	V3 s = primaryTarget(camera, width, height, x, y, aa_x, aa_y, k);
	out[0] = s.x; out[1] = s.y; out[2] = s.z;
}

int oracle_trace_batch(const rt_flat_scene* scene, const float* e, const float* s, size_t n, int max_depth, uint32_t seed,
                       uint8_t* hit, int32_t* obj_id, float* t, float* rgb, uint64_t counts[2], int threads) {
	if (!scene || !e || !s) return -1;
	if (seed == 0) threads = 1;
	std::vector<uint64_t> cn(std::max(threads, 1), 0), cs(std::max(threads, 1), 0);
	parallelFor(n, threads, [&](size_t a, size_t b, int tid) {
		Tracer T; T.S = scene; T.maxDepth = max_depth; T.seed = seed;
		for (size_t i = a; i < b; i++) {
			T.rayIndex = (uint32_t)i;
			V3 colour; Hit h;
			bool ok = T.trace(V3(e + 3 * i), V3(s + 3 * i), colour, 0, 1, &h);
			if (!ok) colour = V3(scene->background);
			if (hit) hit[i] = ok;
			if (obj_id) obj_id[i] = ok ? T.primId(h.prim) : -1;
			if (t) t[i] = ok ? h.t : 0.0f;
			if (rgb) { rgb[3 * i] = colour.x; rgb[3 * i + 1] = colour.y; rgb[3 * i + 2] = colour.z; }
		}
		cn[tid] = T.nNear; cs[tid] = T.nShadow;
	});
	if (counts) for (size_t k = 0; k < cn.size(); k++) { counts[0] += cn[k]; counts[1] += cs[k]; }
	return 0;
}

int oracle_intersect_batch(const rt_flat_scene* scene, const float* e, const float* d, size_t n, uint8_t* hit, int32_t* obj_id, float* t) {
	if (!scene || !e || !d) return -1;
	Tracer T; T.S = scene; T.maxDepth = RT_MAX_REFLECTIONS; T.seed = 1;
	for (size_t i = 0; i < n; i++) {
		Hit h = T.findNearest(V3(e + 3 * i), V3(d + 3 * i));
		bool ok = h.prim != RT_NO_HIT;
		if (hit) hit[i] = ok;
		if (obj_id) obj_id[i] = ok ? T.primId(h.prim) : -1;
		if (t) t[i] = ok ? h.t : 0.0f;
	}
	return 0;
}

int oracle_shadow_batch(const rt_flat_scene* scene, const float* point, const float* light_pos, size_t n, uint8_t* lit, float* shadow) {
	if (!scene || !point || !light_pos) return -1;
	Tracer T; T.S = scene; T.maxDepth = RT_MAX_REFLECTIONS; T.seed = 1;
	for (size_t i = 0; i < n; i++) {
		V3 sh(1.0, 1.0, 1.0);
		bool ok = T.calcShadow(V3(point + 3 * i), V3(light_pos + 3 * i), sh);
		if (lit) lit[i] = ok;
		if (shadow) { shadow[3 * i] = sh.x; shadow[3 * i + 1] = sh.y; shadow[3 * i + 2] = sh.z; }
	}
	return 0;
}

int oracle_render(const rt_flat_scene* scene, const rt_camera* camera, int width, int height, int aa_x, int aa_y, int max_depth, uint32_t seed,
                  float* rgb, uint64_t counts[2], int threads) {
	if (!scene || !camera || !rgb || width <= 0 || height <= 0 || aa_x <= 0 || aa_y <= 0) return -1;
	const int aaSq = aa_x * aa_y;
	if (seed == 0) threads = 1;
	std::vector<uint64_t> cn(std::max(threads, 1), 0), cs(std::max(threads, 1), 0);
	// rows are dealt round-robin so that expensive image regions spread over the threads
	auto rowsOf = [&](int tid, int nthreads) {
		Tracer T; T.S = scene; T.maxDepth = max_depth; T.seed = seed;
		for (int y = tid; y < height; y += nthreads)
			for (int x = 0; x < width; x++) {
				size_t i = (size_t)y * width + x;
				V3 total(0.0, 0.0, 0.0);
				for (int k = 0; k < aaSq; k++) {   // q1.cpp:131-146
					T.rayIndex = (uint32_t)(i * (size_t)aaSq + k);
					V3 colour;
					if (!T.trace(V3(camera->eye), primaryTarget(camera, width, height, x, y, aa_x, aa_y, k), colour, 0, 1, nullptr)) colour = V3(scene->background);
					total = total + colour;
				}
				if (aaSq > 1) total = total / float(aaSq);
				rgb[3 * i] = total.x; rgb[3 * i + 1] = total.y; rgb[3 * i + 2] = total.z;
			}
		cn[tid] = T.nNear; cs[tid] = T.nShadow;
	};
	if (threads <= 1) rowsOf(0, 1);
	else {
		std::vector<std::thread> pool;
		for (int t = 0; t < threads; t++) pool.emplace_back(rowsOf, t, threads);
		for (auto& th : pool) th.join();
	}
	if (counts) for (size_t k = 0; k < cn.size(); k++) { counts[0] += cn[k]; counts[1] += cs[k]; }
	return 0;
}

uint64_t oracle_texel_cases(int reset) {
	const uint64_t n = g_texelCases.load();
	if (reset) g_texelCases.store(0);
	return n;
}

int oracle_texel_state(int on) { return g_texelState.exchange(on ? 1 : 0); }

uint64_t oracle_state_cases(int reset) {
	const uint64_t n = g_stateCases.load();
	if (reset) g_stateCases.store(0);
	return n;
}

}  // extern "C"
