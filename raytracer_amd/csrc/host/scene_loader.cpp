// scene_loader.cpp -- host side: JSON scene -> flat scene IR (include/rt_scene.h).
//
// Restates, over flat arrays, what the reference does at load time:
//   choose_scene            raytracer.cpp:117-305   (schema, defaults, object/light construction)
//   create_csgNode          raytracer.cpp:46-105
//   Sphere/Triangle/Box ctors, Mesh::setBox         objects.cpp:146-157,235-249,315-335
//   csg_node::getBox        csg.cpp:56-85
//   light ctors             objects.cpp:393-397,437-443 ; arealight.cpp:24-35,64-77
//   BVH::BVH / splitNode / BVH_node / sortObjects   bvh.cpp:6-70,162-214
// The tree is built with the reference's algorithm (median split on the longest axis, std::sort on
// centroids, stop at <= 2 objects or depth 16) so that leaf contents and DFS order -- which decide ties
// between equal-t hits -- are the reference's.  libstdc++'s std::sort is used on purpose: its permutation
// for equal keys is part of the oracle's definition ("Ties").
#include "rt_capi.h"
#include "rt_vec.h"
#include "bmp_io.h"
#include "json_min.h"
#include "host_internal.h"
#include "std_sort_par.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <future>
#include <memory>
#include <new>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

using rt::vec3;
using rtjson::Value;

namespace rt {

thread_local std::string g_lastError;
void set_error(const std::string& s) { g_lastError = s; }

void FlatSceneStore::link() {
	rt_flat_scene& s = view;
	s.n_planes = (int32_t)planes.size(); s.planes = planes.data();
	s.n_spheres = (int32_t)spheres.size(); s.spheres = spheres.data();
	s.n_tris = (int32_t)tris.size(); s.tris = tris.data();
	s.n_boxes = (int32_t)boxes.size(); s.boxes = boxes.data();
	s.n_csgs = (int32_t)csgs.size(); s.csgs = csgs.data();
	s.n_csg_nodes = (int32_t)csg_nodes.size(); s.csg_nodes = csg_nodes.data();
	s.n_meshes = (int32_t)meshes.size(); s.meshes = meshes.data();
	s.n_mesh_tris = (int32_t)mesh_tris.size(); s.mesh_tris = mesh_tris.data();
	s.n_uvs = (int32_t)uvs.size(); s.uvs = uvs.data();
	s.n_materials = (int32_t)materials.size(); s.materials = materials.data();
	s.n_lights = (int32_t)lights.size(); s.lights = lights.data();
	s.n_textures = (int32_t)textures.size(); s.textures = textures.data();
	s.n_texel_bytes = (int64_t)texels.size(); s.texels = texels.data();
	s.n_bvh_nodes = (int32_t)bvh_nodes.size(); s.bvh_nodes = bvh_nodes.data();
	s.n_leaf_prims = (int32_t)leaf_prims.size(); s.leaf_prims = leaf_prims.data();
}

}  // namespace rt

namespace {

vec3 to_vec3(const Value& v) { return vec3(v[0].f32(), v[1].f32(), v[2].f32()); }   // vector_to_vec3, raytracer.cpp:38-40
void put3(float* dst, vec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

struct Box6 { float b[6]; };   // minX,maxX,minY,maxY,minZ,maxZ

// f(begin, end, chunk) over [0, n) in up to kMaxChunks chunks of at least minChunk items, each on a thread of its own (the caller's included)
constexpr size_t kMaxChunks = 16;
template <class F> void parallelChunks(size_t n, size_t minChunk, F f) {
	const size_t hw = std::max<size_t>(1, std::min<size_t>(kMaxChunks, std::thread::hardware_concurrency()));
	const size_t chunks = std::min(hw, n / std::max<size_t>(minChunk, 1));
	if (chunks <= 1) { f((size_t)0, n, (size_t)0); return; }
	std::vector<std::future<void>> rest;
	for (size_t c = 1; c < chunks; c++) rest.push_back(std::async(std::launch::async, [=, &f] { f(n * c / chunks, n * (c + 1) / chunks, c); }));
	f((size_t)0, n / chunks, (size_t)0);
	for (auto& r : rest) r.get();
}

// An object as the BVH constructor sees it (bvh.cpp:10-28): its box, its centroid, and where it lives.
struct BuildObj {
	uint32_t type;      // RT_PRIM_*
	int32_t src;        // index into the scene-order staging array of that type
	Box6 box;
	vec3 centroid;
};

// Triangles of a mesh object: the reference's "triangles" (+ "uvCoords") arrays, or -- additive, SURVEY.md section
// 8f-2 -- "file": a little-endian binary next to the scene's assets, so that a 5 M-triangle mesh does not have to go
// through half a gigabyte of JSON text:
//   char magic[8] = "RTMESH01"; uint64 triangles; uint32 has_uv; uint32 reserved;
//   float xyz[triangles][3][3];  float uv[triangles][3][2] (if has_uv)
struct MeshData {
	const Value* tj = nullptr;
	const Value* uvj = nullptr;
	rt::PodVec<float> pos, uv;   // (read straight from the file: no zero-fill first)
	size_t count = 0;
	bool hasUv = false;

	void load(const Value& object, const std::string& assetDir) {
		if (object.has("file")) {
			const std::string path = assetDir + object["file"].string();
			FILE* f = fopen(path.c_str(), "rb");
			if (!f) throw std::runtime_error("mesh file not found: " + path);
			char magic[8]; uint64_t n = 0; uint32_t flags[2] = { 0, 0 };
			bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "RTMESH01", 8) == 0 && fread(&n, 8, 1, f) == 1 && fread(flags, 4, 2, f) == 2 && n < (1ull << 28);
			if (ok) { pos.resize((size_t)n * 9); ok = fread(pos.data(), 4, pos.size(), f) == pos.size(); }
			if (ok && flags[0]) { uv.resize((size_t)n * 6); ok = fread(uv.data(), 4, uv.size(), f) == uv.size(); }
			fclose(f);
			if (!ok) throw std::runtime_error("malformed mesh file: " + path);
			count = (size_t)n; hasUv = flags[0] != 0;
		} else {
			tj = &object["triangles"];
			count = tj->size();
			if (object.has("uvCoords")) { uvj = &object["uvCoords"]; hasUv = true; }
		}
	}
	vec3 vertex(size_t i, int k) const {
		if (tj) return to_vec3((*tj)[i][k]);
		const float* p = pos.data() + i * 9 + k * 3;
		return vec3(p[0], p[1], p[2]);
	}
	void uvOf(size_t i, float out[6]) const {
		if (uvj) { const Value& u = (*uvj)[i]; for (int k = 0; k < 3; k++) { out[2 * k] = u[k][0].f32(); out[2 * k + 1] = u[k][1].f32(); } }
		else if (!uv.empty()) memcpy(out, uv.data() + i * 6, 6 * sizeof(float));
		else throw std::runtime_error("texturemesh without uvCoords");
	}
};

struct Loader {
	rt::FlatSceneStore* S;
	std::string assetDir;
	// scene-order staging (the BVH flatten re-emits these in leaf order)
	std::vector<rt_sphere> spheres;
	rt::PodVec<rt_tri> tris;
	std::vector<rt_box> boxes;
	std::vector<rt_csg> csgs;
	rt::PodVec<BuildObj> bvhObjs;
	int32_t nextBvhId = 0;

	int addTexture(const std::string& file) {
		rt::Image8 img;
		rt::read_bmp(assetDir + file, img);   // failure leaves EasyBMP's 1x1 white image, silently (EasyBMP.cpp:685-695)
		rt_texture t;
		t.width = img.width; t.height = img.height; t.offset = (int64_t)S->texels.size();
		S->texels.insert(S->texels.end(), img.rgb.begin(), img.rgb.end());
		S->textures.push_back(t);
		return (int)S->textures.size() - 1;
	}

	static Box6 triBox(const rt_tri& t) {   // objects.cpp:243-248 (std::min/max over an initializer_list)
		Box6 b;
		for (int a = 0; a < 3; a++) {
			b.b[2 * a] = std::min({ t.p0[a], t.p1[a], t.p2[a] });
			b.b[2 * a + 1] = std::max({ t.p0[a], t.p1[a], t.p2[a] });
		}
		return b;
	}
	static rt_tri makeTri(vec3 p0, vec3 p1, vec3 p2, int mat, int mesh) {   // Triangle::Triangle, objects.cpp:235-249
		rt_tri t;
		put3(t.p0, p0); put3(t.p1, p1); put3(t.p2, p2);
		put3(t.n, rt::normalize(rt::cross(p1 - p0, p2 - p1)));
		t.id = -1; t.mat = mat; t.mesh = mesh; t.uv = -1;
		return t;
	}
	static Box6 boxFromPoints(vec3 p1, vec3 p2) {   // raytracer.cpp:71-79,244-252
		Box6 b;
		b.b[0] = std::min(p1.x, p2.x); b.b[1] = std::max(p1.x, p2.x);
		b.b[2] = std::min(p1.y, p2.y); b.b[3] = std::max(p1.y, p2.y);
		b.b[4] = std::min(p1.z, p2.z); b.b[5] = std::max(p1.z, p2.z);
		return b;
	}
	static Box6 sphereBox(vec3 c, float r) {   // objects.cpp:151-156
		Box6 b;
		b.b[0] = c.x - r; b.b[1] = c.x + r; b.b[2] = c.y - r; b.b[3] = c.y + r; b.b[4] = c.z - r; b.b[5] = c.z + r;
		return b;
	}
	static Box6 meshBox(const rt::PodVec<rt_tri>& ts, size_t first, size_t count) {   // Mesh::setBox, objects.cpp:315-325
		Box6 b = triBox(ts[first]);
		for (size_t i = 1; i < count; i++) {
			Box6 c = triBox(ts[first + i]);
			b.b[0] = std::min(b.b[0], c.b[0]); b.b[2] = std::min(b.b[2], c.b[2]); b.b[4] = std::min(b.b[4], c.b[4]);
			b.b[1] = std::max(b.b[1], c.b[1]); b.b[3] = std::max(b.b[3], c.b[3]); b.b[5] = std::max(b.b[5], c.b[5]);
		}
		return b;
	}

	// create_csgNode (raytracer.cpp:46-105) + csg_node::getBox (csg.cpp:56-85); emits nodes in post-order.
	Box6 addCsgNode(const Value& nj, int& leaves) {
		rt_csg_node node;
		memset(&node, 0, sizeof node);
		node.mesh = -1;
		Box6 box;
		memset(&box, 0, sizeof box);
		if (nj.has("operation")) {
			const Value& op = nj["operation"];
			if (op.is("union")) node.kind = RT_CSG_UNION;
			else if (op.is("intersection")) node.kind = RT_CSG_INTERSECTION;
			else if (op.is("difference")) node.kind = RT_CSG_DIFFERENCE;
			else throw std::runtime_error("csg: unknown operation");
			Box6 b1 = addCsgNode(nj["first"], leaves);
			Box6 b2 = addCsgNode(nj["second"], leaves);
			if (node.kind == RT_CSG_DIFFERENCE) box = b1;
			else if (node.kind == RT_CSG_UNION) {
				for (int a = 0; a < 3; a++) { box.b[2 * a] = std::min(b1.b[2 * a], b2.b[2 * a]); box.b[2 * a + 1] = std::max(b1.b[2 * a + 1], b2.b[2 * a + 1]); }
			} else {
				for (int a = 0; a < 3; a++) { box.b[2 * a] = std::max(b1.b[2 * a], b2.b[2 * a]); box.b[2 * a + 1] = std::min(b1.b[2 * a + 1], b2.b[2 * a + 1]); }
			}
		} else {
			leaves++;
			const Value& type = nj["type"];
			if (type.is("sphere")) {
				vec3 c = to_vec3(nj["position"]);
				float r = nj["radius"].f32();
				node.kind = RT_CSG_SPHERE;
				node.f[0] = c.x; node.f[1] = c.y; node.f[2] = c.z; node.f[3] = r;
				box = sphereBox(c, r);
			} else if (type.is("box")) {
				box = boxFromPoints(to_vec3(nj["point1"]), to_vec3(nj["point2"]));
				node.kind = RT_CSG_BOX;
				memcpy(node.f, box.b, sizeof box.b);
			} else if (type.is("mesh")) {
				MeshData md;
				md.load(nj, assetDir);
				rt_mesh m;
				m.tri_first = (int32_t)S->mesh_tris.size(); m.tri_count = (int32_t)md.count; m.tex = -1; m.mat = 0;
				if (m.tri_count == 0) throw std::runtime_error("csg: mesh leaf without triangles");
				int meshIdx = (int)S->meshes.size();
				for (size_t i = 0; i < md.count; i++)
					S->mesh_tris.push_back(makeTri(md.vertex(i, 0), md.vertex(i, 1), md.vertex(i, 2), 0, meshIdx));
				S->meshes.push_back(m);
				node.kind = RT_CSG_MESH; node.mesh = meshIdx;
				box = meshBox(S->mesh_tris, (size_t)m.tri_first, (size_t)m.tri_count);
			} else throw std::runtime_error("csg: unknown leaf type");
		}
		S->csg_nodes.push_back(node);
		return box;
	}

	// ---- BVH (bvh.cpp:37-70,162-214) ----
	static void growBox(Box6& b, const Box6& c) {
		b.b[0] = std::min(b.b[0], c.b[0]); b.b[1] = std::max(b.b[1], c.b[1]);
		b.b[2] = std::min(b.b[2], c.b[2]); b.b[3] = std::max(b.b[3], c.b[3]);
		b.b[4] = std::min(b.b[4], c.b[4]); b.b[5] = std::max(b.b[5], c.b[5]);
	}
	// (min and max are exact and associative: the chunks of a long list may be reduced on separate threads)
	static Box6 unionBox(const rt::PodVec<Box6>& all, const int* objs, size_t n) {   // BVH_node::BVH_node
		if (n < 200000) {
			Box6 b = all[(size_t)objs[0]];
			for (size_t i = 1; i < n; i++) growBox(b, all[(size_t)objs[i]]);
			return b;
		}
		std::vector<Box6> part(kMaxChunks, all[(size_t)objs[0]]);
		parallelChunks(n, 50000, [&](size_t begin, size_t end, size_t chunk) {
			Box6 b = all[(size_t)objs[begin]];
			for (size_t i = begin + 1; i < end; i++) growBox(b, all[(size_t)objs[i]]);
			part[chunk] = b;
		});
		Box6 b = part[0];
		for (const Box6& c : part) growBox(b, c);
		return b;
	}

	// The reference's build (bvh.cpp:37-70) in two steps: the splits -- each node sorts its own object list, so sibling
	// subtrees are independent and the top levels are built on separate threads (the sorts are where the time of a
	// 5 M-triangle scene goes; every sort sees exactly the list it sees in the reference, so the tree is the same) --
	// and then the flattening into pre-order nodes and leaf-order primitives, every subtree into ranges known from the counts.
	struct TreeNode {
		Box6 lb, rb;
		std::unique_ptr<TreeNode> l, r;
		size_t first = 0, count = 0;   // leaf contents (leaf iff l == nullptr): order[first, first + count)
		// what the subtree adds to the flat arrays: inner nodes, and primitives of each type (spheres, triangles, boxes, CSG objects)
		int64_t nNodes = 0, nOfType[4] = { 0, 0, 0, 0 };
		int64_t prims() const { return nOfType[0] + nOfType[1] + nOfType[2] + nOfType[3]; }
	};
	// A node's object list is a range of `order` (indices into bvhObjs): sorting the range and halving it leaves the two children's
	// lists where they are, in the order the reference's copies would have (bvh.cpp:58-66), and no list is ever allocated.  `keyed` is
	// the scratch the sorts run in: the range's centroids along the split axis gathered beside the indices.
	struct KeyIdx { float key; int idx; };
	rt::PodVec<int> order;
	rt::PodVec<KeyIdx> keyed;
	rt::PodVec<float> centroidOf[3];   // the objects' centroids, one array per axis, and their boxes: what the splits read, 4 and 24 B per object
	rt::PodVec<Box6> boxOf;
	std::unique_ptr<TreeNode> splitNode(size_t first, size_t n, const Box6& box, int depth) {
		std::unique_ptr<TreeNode> t(new TreeNode());
		if (n <= 2 || depth >= RT_MAX_BVH_DEPTH) {   // bvh.cpp:39
			t->first = first; t->count = n;
			for (size_t i = 0; i < n; i++) t->nOfType[bvhObjs[(size_t)order[first + i]].type]++;
			return t;
		}
		float lengthX = box.b[1] - box.b[0], lengthY = box.b[3] - box.b[2], lengthZ = box.b[5] - box.b[4];
		int axis = (lengthX >= lengthY && lengthX >= lengthZ) ? 0 : (lengthY >= lengthZ ? 1 : 2);   // bvh.cpp:48-56
		const float* cen = centroidOf[axis].data();
		// sortObjects (bvh.cpp:188-213): std::sort of the list by centroid.  With the keys beside the indices the comparisons, and with them
		// the permutation, stay the ones of std::sort(objs, [](a, b) { return centroid(a) < centroid(b); }) without two random reads per
		// comparison; a long list is sorted by std_sort_par.h, which is std::sort on several threads.
		int* objs = order.data() + first;
		KeyIdx* keys = keyed.data() + first;
		bool ordered = true;   // (a NaN centroid is no strict weak ordering: whatever std::sort does with it, it does here)
		if (n >= 200000) {
			std::vector<char> bad(kMaxChunks, 0);
			parallelChunks(n, 50000, [&](size_t begin, size_t end, size_t chunk) {
				for (size_t i = begin; i < end; i++) { keys[i] = { cen[(size_t)objs[i]], objs[i] }; if (keys[i].key != keys[i].key) bad[chunk] = 1; }
			});
			for (char c : bad) if (c) ordered = false;
		} else for (size_t i = 0; i < n; i++) { keys[i] = { cen[(size_t)objs[i]], objs[i] }; if (keys[i].key != keys[i].key) ordered = false; }
		auto less = [](const KeyIdx& x, const KeyIdx& y) { return x.key < y.key; };
		static const bool plainSort = getenv("RT_LOADER_STD_SORT") != nullptr;   // (the switch: for the tests)
		if (n >= 100000 && ordered && !plainSort) rt::stdSortParallel(keys, (long)n, less, (long)std::max<size_t>(1 << 15, n / 64));
		else std::sort(keys, keys + n, less);
		for (size_t i = 0; i < n; i++) objs[i] = keys[i].idx;
		const size_t nl = n / 2, nr = n - n / 2;   // bvh.cpp:58-66
		t->lb = unionBox(boxOf, objs, nl); t->rb = unionBox(boxOf, objs + nl, nr);
		if (depth < 6 && n > 20000) {   // (64 subtrees at most: enough to keep 16 cores busy to the end)
			auto fut = std::async(std::launch::async, [&] { return splitNode(first, nl, t->lb, depth + 1); });
			t->r = splitNode(first + nl, nr, t->rb, depth + 1);
			t->l = fut.get();
		} else {
			t->l = splitNode(first, nl, t->lb, depth + 1);
			t->r = splitNode(first + nl, nr, t->rb, depth + 1);
		}
		t->nNodes = 1 + t->l->nNodes + t->r->nNodes;
		for (int k = 0; k < 4; k++) t->nOfType[k] = t->l->nOfType[k] + t->r->nOfType[k];
		return t;
	}
	// Where a subtree's output starts in the flat arrays: its inner nodes in bvh_nodes (pre-order), its primitives in leaf_prims (leaf
	// order) and in the arrays of their types.  The counts of every subtree are known from the split (TreeNode::nNodes / nOfType), so the
	// subtrees of the top levels are written on separate threads, each into its own ranges -- the arrays come out as a sequential
	// pre-order walk would fill them.
	struct FlatAt { int64_t node, prim, ofType[4]; };
	// Returns the (child, count) reference of the subtree `t`.
	void flattenNode(const TreeNode& t, FlatAt at, int depth, int32_t& childOut, int32_t& countOut) {
		if (!t.l) {
			childOut = ~(int32_t)at.prim;
			countOut = (int32_t)t.count;
			for (size_t i = 0; i < t.count; i++) {
				const BuildObj& o = bvhObjs[(size_t)order[t.first + i]];
				const int64_t k = at.ofType[o.type]++;
				S->leaf_prims[(size_t)at.prim++] = RT_PRIM_REF(o.type, k);
				switch (o.type) {
				case RT_PRIM_SPHERE: S->spheres[(size_t)k] = spheres[(size_t)o.src]; break;
				case RT_PRIM_TRI: S->tris[(size_t)k] = tris[(size_t)o.src]; break;
				case RT_PRIM_BOX: S->boxes[(size_t)k] = boxes[(size_t)o.src]; break;
				case RT_PRIM_CSG: S->csgs[(size_t)k] = csgs[(size_t)o.src]; break;
				}
			}
			S->leaf_prims[(size_t)at.prim - 1] |= RT_PRIM_LAST;   // device leaf loops run until this flag instead of carrying a count
			return;
		}
		const int64_t self = at.node;
		FlatAt left = at, right = at;
		left.node = self + 1;
		right.node = self + 1 + t.l->nNodes;
		right.prim = at.prim + t.l->prims();
		for (int k = 0; k < 4; k++) right.ofType[k] = at.ofType[k] + t.l->nOfType[k];
		int32_t c0, n0, c1, n1;
		if (depth < 4 && t.prims() > 20000) {
			auto fut = std::async(std::launch::async, [&] { flattenNode(*t.l, left, depth + 1, c0, n0); });
			flattenNode(*t.r, right, depth + 1, c1, n1);
			fut.get();
		} else {
			flattenNode(*t.l, left, depth + 1, c0, n0);
			flattenNode(*t.r, right, depth + 1, c1, n1);
		}
		rt_bvh_node& n = S->bvh_nodes[(size_t)self];
		memcpy(n.box[0], t.lb.b, sizeof t.lb.b); memcpy(n.box[1], t.rb.b, sizeof t.rb.b);
		n.child[0] = c0; n.child[1] = c1; n.count[0] = n0; n.count[1] = n1;
		childOut = (int32_t)self; countOut = 0;
	}
	void flattenTree(const TreeNode& root, int32_t& childOut, int32_t& countOut) {
		S->bvh_nodes.assign((size_t)root.nNodes, rt_bvh_node());
		S->leaf_prims.assign((size_t)root.prims(), 0u);
		S->spheres.resize((size_t)root.nOfType[RT_PRIM_SPHERE]); S->tris.resize((size_t)root.nOfType[RT_PRIM_TRI]);
		S->boxes.resize((size_t)root.nOfType[RT_PRIM_BOX]); S->csgs.resize((size_t)root.nOfType[RT_PRIM_CSG]);
		flattenNode(root, FlatAt{ 0, 0, { 0, 0, 0, 0 } }, 0, childOut, countOut);
	}

	static rt_material parseMaterial(const Value& mj) {   // raytracer.cpp:149-165
		rt_material m;
		memset(&m, 0, sizeof m);
		if (mj.has("ambient")) put3(m.ambient, to_vec3(mj["ambient"]));
		if (mj.has("diffuse")) put3(m.diffuse, to_vec3(mj["diffuse"]));
		if (mj.has("specular")) put3(m.specular, to_vec3(mj["specular"]));
		if (mj.has("reflective")) put3(m.reflective, to_vec3(mj["reflective"]));
		if (mj.has("transmissive")) put3(m.transmissive, to_vec3(mj["transmissive"]));
		if (mj.has("shininess")) m.shininess = mj["shininess"].f32();
		if (mj.has("refraction")) m.refraction = mj["refraction"].f32();
		return m;
	}

	void load(const Value& scene) {
		rt_flat_scene& V = S->view;
		V.fov = 60; V.background[0] = V.background[1] = V.background[2] = 0;   // raytracer.cpp:24-25
		const Value& camera = scene["camera"];
		if (camera.has("field")) { V.fov = camera["field"].number(); S->hasCameraField = true; }
		if (camera.has("background")) { put3(V.background, to_vec3(camera["background"])); S->hasCameraBackground = true; }

		// material 0 = Material() for csg leaves; object k uses material k+1
		rt_material zero; memset(&zero, 0, sizeof zero);
		S->materials.push_back(zero);

		const Value& objects = scene["objects"];
		// ids: planes first, then BVH objects in scene order (bvh.cpp:10-28)
		int nPlanes = 0;
		for (size_t i = 0; i < objects.size(); i++) if (objects[i]["type"].is("plane")) nPlanes++;
		int planeId = 0;
		nextBvhId = nPlanes;

		for (size_t oi = 0; oi < objects.size(); oi++) {
			const Value& object = objects[oi];
			int mat = (int)S->materials.size();
			S->materials.push_back(parseMaterial(object["material"]));
			const Value& type = object["type"];
			if (type.is("sphere") || type.is("bumpsphere")) {
				rt_sphere s;
				vec3 c = to_vec3(object["position"]);
				put3(s.center, c); s.radius = object["radius"].f32();
				s.id = nextBvhId++; s.mat = mat; s.bump_tex = -1; s.bump_depth = 0;
				if (type.is("bumpsphere")) {   // raytracer.cpp:226-233
					std::string file = object["bumpmap"].string();
					s.bump_depth = object["bumpdepth"].f32();
					s.bump_tex = addTexture(file);
				}
				BuildObj b; b.type = RT_PRIM_SPHERE; b.src = (int)spheres.size(); b.box = sphereBox(c, s.radius); b.centroid = c;
				spheres.push_back(s); bvhObjs.push_back(b);
			} else if (type.is("plane")) {
				rt_plane p;
				put3(p.point, to_vec3(object["position"])); put3(p.normal, to_vec3(object["normal"]));
				p.id = planeId++; p.mat = mat;
				S->planes.push_back(p);
			} else if (type.is("mesh") || type.is("texturemesh")) {
				const bool timing = getenv("RT_LOAD_TIMING") != nullptr;
				auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
				const double m0 = now();
				MeshData md;
				md.load(object, assetDir);
				const double m1 = now();
				bool textured = type.is("texturemesh");
				rt_mesh m;
				m.tri_first = (int32_t)S->mesh_tris.size(); m.tri_count = (int32_t)md.count; m.tex = -1; m.mat = mat;
				int meshIdx = (int)S->meshes.size();
				if (textured) m.tex = addTexture(object["texture"].string());
				// (every triangle's records land at known places: the triangles of a large mesh are made on several threads)
				const size_t triBase = tris.size(), meshTriBase = S->mesh_tris.size(), objBase = bvhObjs.size(), uvBase = S->uvs.size();
				const int32_t idBase = nextBvhId;
				tris.resize(triBase + md.count); S->mesh_tris.resize(meshTriBase + md.count); bvhObjs.resize(objBase + md.count);
				if (textured) S->uvs.resize(uvBase + md.count);
				parallelChunks(md.count, 100000, [&](size_t begin, size_t end, size_t) {
					for (size_t i = begin; i < end; i++) {
						rt_tri t = makeTri(md.vertex(i, 0), md.vertex(i, 1), md.vertex(i, 2), mat, meshIdx);
						t.id = idBase + (int32_t)i;
						if (textured) {   // raytracer.cpp:210-220
							rt_uv uv;
							md.uvOf(i, uv.uv);
							t.uv = (int32_t)(uvBase + i);
							S->uvs[uvBase + i] = uv;
						}
						BuildObj b; b.type = RT_PRIM_TRI; b.src = (int)(triBase + i); b.box = triBox(t);
						b.centroid = vec3((t.p0[0] + t.p1[0] + t.p2[0]) / 3, (t.p0[1] + t.p1[1] + t.p2[1]) / 3, (t.p0[2] + t.p1[2] + t.p2[2]) / 3);   // objects.cpp:267-271
						tris[triBase + i] = t; S->mesh_tris[meshTriBase + i] = t; bvhObjs[objBase + i] = b;
					}
				});
				nextBvhId += (int32_t)md.count;
				S->meshes.push_back(m);
				if (timing && md.count > 100000) fprintf(stderr, "[load] mesh of %zu triangles: read %.2f s, triangles %.2f s\n", md.count, m1 - m0, now() - m1);
			} else if (type.is("csgobject")) {
				rt_csg c;
				memset(&c, 0, sizeof c);
				c.id = nextBvhId++; c.mat = mat; c.node_first = (int32_t)S->csg_nodes.size();
				if (!object.has("operation")) throw std::runtime_error("csgobject without an operation");
				int leaves = 0;
				Box6 box = addCsgNode(object, leaves);
				if (leaves > RT_MAX_CSG_LEAVES) throw std::runtime_error("csgobject has more leaves than RT_MAX_CSG_LEAVES");
				c.node_count = (int32_t)S->csg_nodes.size() - c.node_first;
				memcpy(c.b, box.b, sizeof box.b);
				BuildObj b; b.type = RT_PRIM_CSG; b.src = (int)csgs.size(); b.box = box;
				b.centroid = vec3((box.b[0] + box.b[1]) / 2, (box.b[2] + box.b[3]) / 2, (box.b[4] + box.b[5]) / 2);   // csg.cpp:32-36
				csgs.push_back(c); bvhObjs.push_back(b);
			} else if (type.is("box")) {
				rt_box bx;
				Box6 box = boxFromPoints(to_vec3(object["point1"]), to_vec3(object["point2"]));
				memcpy(bx.b, box.b, sizeof box.b); bx.id = nextBvhId++; bx.mat = mat;
				BuildObj b; b.type = RT_PRIM_BOX; b.src = (int)boxes.size(); b.box = box;
				b.centroid = vec3((box.b[0] + box.b[1]) / 2, (box.b[2] + box.b[3]) / 2, (box.b[4] + box.b[5]) / 2);   // objects.cpp:363-367
				boxes.push_back(bx); bvhObjs.push_back(b);
			}
			// unknown types are ignored, as in the reference's chain of ifs
		}
		V.n_objects = nextBvhId;

		const Value& lights = scene["lights"];
		for (size_t li = 0; li < lights.size(); li++) {
			const Value& lj = lights[li];
			rt_light L;
			memset(&L, 0, sizeof L);
			put3(L.colour, to_vec3(lj["color"]));
			const Value& type = lj["type"];
			if (type.is("ambient")) L.type = RT_LIGHT_AMBIENT;
			else if (type.is("directional")) { L.type = RT_LIGHT_DIRECTIONAL; put3(L.direction, rt::normalize(to_vec3(lj["direction"]))); }
			else if (type.is("point")) { L.type = RT_LIGHT_POINT; put3(L.position, to_vec3(lj["position"])); }
			else if (type.is("spot")) {
				L.type = RT_LIGHT_SPOT;
				put3(L.position, to_vec3(lj["position"]));
				put3(L.direction, rt::normalize(to_vec3(lj["direction"])));
				L.cutoff = lj["cutoff"].f32();
				L.cos_cutoff = cos(L.cutoff * 3.14159265358979323846264338327950288 / 180);   // objects.cpp:451 (objects.h:9)
			} else if (type.is("rectangular")) {   // arealight.cpp:24-35
				L.type = RT_LIGHT_RECT;
				vec3 normal = to_vec3(lj["normal"]), orientation = to_vec3(lj["orientation"]);
				put3(L.position, to_vec3(lj["position"])); put3(L.normal, normal);
				L.width = lj["width"].f32(); L.height = lj["height"].f32(); L.samples = lj["samples"].i32();
				vec3 planeX = rt::normalize(rt::cross(orientation, normal));
				vec3 planeY = rt::normalize(rt::cross(normal, planeX));
				put3(L.planeX, planeX); put3(L.planeY, planeY);
			} else if (type.is("circular")) {   // arealight.cpp:64-77
				L.type = RT_LIGHT_CIRC;
				vec3 normal = to_vec3(lj["normal"]);
				put3(L.position, to_vec3(lj["position"])); put3(L.normal, normal);
				L.radius = lj["radius"].f32(); L.samples = lj["samples"].i32();
				vec3 planeX = rt::normalize(rt::cross(vec3(0, 1, 0), normal));
				if (std::isnan(planeX.x)) planeX = rt::normalize(rt::cross(vec3(0, 0, 1), normal));
				vec3 planeY = rt::normalize(rt::cross(normal, planeX));
				put3(L.planeX, planeX); put3(L.planeY, planeY);
			} else continue;
			S->lights.push_back(L);
		}

		// bvh = new BVH(Objects)  (raytracer.cpp:304).  An all-plane scene dereferences objects[0] in the
		// reference (bvh.cpp:164) and crashes; here it yields an empty tree.
		V.root_child = -1; V.root_count = 0;
		memset(V.root_box, 0, sizeof V.root_box);
		if (!bvhObjs.empty()) {
			order.resize(bvhObjs.size());
			parallelChunks(order.size(), 100000, [&](size_t begin, size_t end, size_t) { for (size_t i = begin; i < end; i++) order[i] = (int)i; });
			keyed.resize(order.size());
			boxOf.resize(order.size());
			for (int a = 0; a < 3; a++) centroidOf[a].resize(order.size());
			parallelChunks(order.size(), 100000, [&](size_t begin, size_t end, size_t) {
				for (size_t i = begin; i < end; i++) {
					const BuildObj& o = bvhObjs[i];
					boxOf[i] = o.box; centroidOf[0][i] = o.centroid.x; centroidOf[1][i] = o.centroid.y; centroidOf[2][i] = o.centroid.z;
				}
			});
			Box6 rootBox = unionBox(boxOf, order.data(), order.size());
			memcpy(V.root_box, rootBox.b, sizeof rootBox.b);
			const bool timing = getenv("RT_LOAD_TIMING") != nullptr;
			const auto t0 = std::chrono::steady_clock::now();
			std::unique_ptr<TreeNode> tree = splitNode(0, order.size(), rootBox, 0);
			const auto t1 = std::chrono::steady_clock::now();
			flattenTree(*tree, V.root_child, V.root_count);
			const auto t2 = std::chrono::steady_clock::now();
			if (timing) fprintf(stderr, "[load] bvh split %.2f s, flatten %.2f s\n", std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
		}
		S->link();
	}
};

int loadFromText(const std::string& text, const char* assetDir, rt_flat_scene** out) {
	if (!out) { rt::set_error("out is NULL"); return RT_EINVAL; }
	*out = nullptr;
	auto* store = new rt::FlatSceneStore();
	try {
		const bool timing = getenv("RT_LOAD_TIMING") != nullptr;
		auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		const double t0 = now();
		Value scene = rtjson::Parser::parse(text);
		const double t1 = now();
		double t2;
		{
			Loader L;
			L.S = store;
			L.assetDir = assetDir ? assetDir : "";
			if (!L.assetDir.empty() && L.assetDir.back() != '/') L.assetDir += '/';
			L.load(scene);
			t2 = now();
		}
		if (timing) fprintf(stderr, "[load] json %.2f s, objects + tree %.2f s, staging freed %.2f s\n", t1 - t0, t2 - t1, now() - t2);
	} catch (const std::exception& e) {
		delete store;
		rt::set_error(std::string("scene: ") + e.what());
		return RT_EPARSE;
	}
	*out = &store->view;
	return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return rt::g_lastError.c_str(); }
const char* rt_version(void) { return "raytracer_amd 0.1 (gfx950)"; }

int rt_load_scene_json_text(const char* json_text, size_t len, const char* asset_dir, rt_flat_scene** out) {
	if (!json_text) { rt::set_error("json_text is NULL"); return RT_EINVAL; }
	return loadFromText(std::string(json_text, len), asset_dir, out);
}

int rt_load_scene_json(const char* json_path, const char* asset_dir, rt_flat_scene** out) {
	if (!json_path) { rt::set_error("json_path is NULL"); return RT_EINVAL; }
	std::ifstream in(json_path, std::ios::binary);
	if (!in.is_open()) { rt::set_error(std::string("Unable to open scene file ") + json_path); return RT_EIO; }
	std::stringstream ss;
	ss << in.rdbuf();
	std::string dir;
	if (asset_dir) dir = asset_dir;
	else { std::string p(json_path); size_t k = p.find_last_of('/'); dir = k == std::string::npos ? "" : p.substr(0, k + 1); }
	return loadFromText(ss.str(), dir.c_str(), out);
}

void rt_flat_scene_free(rt_flat_scene* scene) {
	if (scene) delete rt::FlatSceneStore::from_view(scene);
}

}  // extern "C"

// ---- binary form -------------------------------------------------------------------------------------
static const char kMagic[8] = { 'R', 'T', 'S', 'C', '0', '0', '0', '3' };

template <class T> static bool wr(FILE* f, const T* p, int64_t count) {
	uint64_t n = count > 0 ? (uint64_t)count : 0;
	if (fwrite(&n, 8, 1, f) != 1) return false;
	return n == 0 || (p && fwrite(p, sizeof(T), n, f) == n);
}
// One array of the file.  The count is not trusted: it must fit what is left of the file (a corrupt count must not turn
// into a 64 GB resize) and, except for the texel bytes, the int32 counts of rt_flat_scene.
template <class V> static bool rd(FILE* f, V& v, uint64_t& left, bool wide = false) {
	using T = typename V::value_type;
	uint64_t n = 0;
	if (left < 8 || fread(&n, 8, 1, f) != 1) return false;
	left -= 8;
	if (n > left / sizeof(T) || (!wide && n > 0x7fffffffull)) return false;
	v.resize(n);
	left -= n * sizeof(T);
	return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

extern "C" {

int rt_flat_scene_save(const rt_flat_scene* scene, const char* path) {
	if (!scene || !path) { rt::set_error("NULL argument"); return RT_EINVAL; }
	FILE* f = fopen(path, "wb");
	if (!f) { rt::set_error(std::string("cannot write ") + path); return RT_EIO; }
	const rt_flat_scene& V = *scene;
	bool ok = fwrite(kMagic, 8, 1, f) == 1 && fwrite(&V.fov, 8, 1, f) == 1 && fwrite(V.background, 12, 1, f) == 1 &&
	          fwrite(&V.n_objects, 4, 1, f) == 1 && fwrite(V.root_box, 24, 1, f) == 1 && fwrite(&V.root_child, 4, 1, f) == 1 &&
	          fwrite(&V.root_count, 4, 1, f) == 1;
	ok = ok && wr(f, V.planes, V.n_planes) && wr(f, V.spheres, V.n_spheres) && wr(f, V.tris, V.n_tris) && wr(f, V.boxes, V.n_boxes) &&
	     wr(f, V.csgs, V.n_csgs) && wr(f, V.csg_nodes, V.n_csg_nodes) && wr(f, V.meshes, V.n_meshes) && wr(f, V.mesh_tris, V.n_mesh_tris) &&
	     wr(f, V.uvs, V.n_uvs) && wr(f, V.materials, V.n_materials) && wr(f, V.lights, V.n_lights) && wr(f, V.textures, V.n_textures) &&
	     wr(f, V.texels, V.n_texel_bytes) && wr(f, V.bvh_nodes, V.n_bvh_nodes) && wr(f, V.leaf_prims, V.n_leaf_prims);
	fclose(f);
	if (!ok) { rt::set_error(std::string("short write to ") + path); return RT_EIO; }
	return RT_OK;
}

int rt_flat_scene_load(const char* path, rt_flat_scene** out) {
	if (!out || !path) { rt::set_error("NULL argument"); return RT_EINVAL; }
	*out = nullptr;
	FILE* f = fopen(path, "rb");
	if (!f) { rt::set_error(std::string("Unable to open scene file ") + path); return RT_EIO; }
	uint64_t left = 0;
	if (fseek(f, 0, SEEK_END) == 0) { const long end = ftell(f); if (end > 0) left = (uint64_t)end; }
	rewind(f);
	const uint64_t header = 8 + 8 + 12 + 4 + 24 + 4 + 4;
	rt::FlatSceneStore* S = nullptr;
	bool ok = false;
	try {   // nothing may escape an extern "C" entry point (std::bad_alloc from a resize would abort a ctypes caller)
		S = new rt::FlatSceneStore();
		rt_flat_scene& V = S->view;
		char magic[8];
		ok = left >= header && fread(magic, 8, 1, f) == 1 && !memcmp(magic, kMagic, 8) && fread(&V.fov, 8, 1, f) == 1 && fread(V.background, 12, 1, f) == 1 &&
		     fread(&V.n_objects, 4, 1, f) == 1 && fread(V.root_box, 24, 1, f) == 1 && fread(&V.root_child, 4, 1, f) == 1 &&
		     fread(&V.root_count, 4, 1, f) == 1;
		left -= ok ? header : 0;
		ok = ok && rd(f, S->planes, left) && rd(f, S->spheres, left) && rd(f, S->tris, left) && rd(f, S->boxes, left) && rd(f, S->csgs, left) &&
		     rd(f, S->csg_nodes, left) && rd(f, S->meshes, left) && rd(f, S->mesh_tris, left) && rd(f, S->uvs, left) && rd(f, S->materials, left) &&
		     rd(f, S->lights, left) && rd(f, S->textures, left) && rd(f, S->texels, left, true) && rd(f, S->bvh_nodes, left) && rd(f, S->leaf_prims, left);
		if (ok) S->link();
	} catch (const std::bad_alloc&) {
		fclose(f);
		delete S;
		rt::set_error(std::string("out of memory reading ") + path);
		return RT_ENOMEM;
	} catch (const std::exception& e) {
		fclose(f);
		delete S;
		rt::set_error(std::string("scene file: ") + e.what());
		return RT_EPARSE;
	}
	fclose(f);
	if (!ok) { delete S; rt::set_error(std::string("not a flat scene file, truncated, or with counts beyond the file's size: ") + path); return RT_EPARSE; }
	*out = &S->view;
	return RT_OK;
}

}  // extern "C"
