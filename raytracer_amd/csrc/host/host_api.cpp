// host_api.cpp -- host-side pieces of the C-ABI that are not the loader: scene validation, the camera basis
// (q1.cpp:30-44), RGB8 quantisation and BMP output.
#include "rt_capi.h"
#include "rt_vec.h"
#include "bmp_io.h"
#include "host_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <vector>

using rt::vec3;

namespace rt {

#define RT_CHECK(cond, msg) do { if (!(cond)) { set_error(std::string("invalid scene: ") + msg); return RT_EINVAL; } } while (0)

static bool childOk(const rt_flat_scene* s, int32_t child, int32_t count) {
	if (child >= 0) return child < s->n_bvh_nodes && count == 0;
	int64_t first = ~(int64_t)child;
	if (count < 1 || first + count > s->n_leaf_prims) return false;
	// the device walks a leaf until it meets RT_PRIM_LAST: exactly the last entry of the run must carry it
	for (int64_t k = 0; k < count; k++)
		if (((s->leaf_prims[first + k] & RT_PRIM_LAST) != 0) != (k == count - 1)) return false;
	return true;
}

int validate_flat_scene(const rt_flat_scene* s) {
	RT_CHECK(s, "NULL scene");
	RT_CHECK(s->n_planes >= 0 && s->n_spheres >= 0 && s->n_tris >= 0 && s->n_boxes >= 0 && s->n_csgs >= 0 && s->n_csg_nodes >= 0 &&
	         s->n_meshes >= 0 && s->n_mesh_tris >= 0 && s->n_uvs >= 0 && s->n_materials >= 1 && s->n_lights >= 0 && s->n_textures >= 0 &&
	         s->n_bvh_nodes >= 0 && s->n_leaf_prims >= 0 && s->n_texel_bytes >= 0, "negative count or no materials");
	RT_CHECK(s->n_tris < (1 << 28) && s->n_spheres < (1 << 28) && s->n_boxes < (1 << 28) && s->n_csgs < (1 << 28), "too many primitives");
	auto matOk = [&](int m) { return m >= 0 && m < s->n_materials; };
	for (int i = 0; i < s->n_planes; i++) RT_CHECK(matOk(s->planes[i].mat), "plane material");
	for (int i = 0; i < s->n_spheres; i++) {
		RT_CHECK(matOk(s->spheres[i].mat), "sphere material");
		RT_CHECK(s->spheres[i].bump_tex < s->n_textures, "sphere bump texture");
	}
	for (int i = 0; i < s->n_boxes; i++) RT_CHECK(matOk(s->boxes[i].mat), "box material");
	for (int pass = 0; pass < 2; pass++) {
		const rt_tri* ts = pass ? s->mesh_tris : s->tris;
		int n = pass ? s->n_mesh_tris : s->n_tris;
		for (int i = 0; i < n; i++) {
			RT_CHECK(matOk(ts[i].mat), "triangle material");
			RT_CHECK(ts[i].mesh >= 0 && ts[i].mesh < s->n_meshes, "triangle mesh index");
			RT_CHECK(ts[i].uv < s->n_uvs, "triangle uv index");
			RT_CHECK(ts[i].uv < 0 || s->meshes[ts[i].mesh].tex >= 0, "uv on an untextured mesh");
		}
	}
	for (int i = 0; i < s->n_meshes; i++) {
		const rt_mesh& m = s->meshes[i];
		RT_CHECK(m.tri_first >= 0 && m.tri_count >= 0 && (int64_t)m.tri_first + m.tri_count <= s->n_mesh_tris, "mesh triangle range");
		RT_CHECK(m.tex < s->n_textures && matOk(m.mat), "mesh texture/material");
	}
	for (int i = 0; i < s->n_textures; i++) {
		const rt_texture& t = s->textures[i];
		RT_CHECK(t.width >= 1 && t.height >= 1 && t.offset >= 0 && t.offset + (int64_t)t.width * t.height * 3 <= s->n_texel_bytes, "texture range");
	}
	for (int i = 0; i < s->n_csgs; i++) {
		const rt_csg& c = s->csgs[i];
		RT_CHECK(matOk(c.mat), "csg material");
		RT_CHECK(c.node_first >= 0 && c.node_count >= 1 && (int64_t)c.node_first + c.node_count <= s->n_csg_nodes, "csg node range");
		// the post-order program must be a well-formed expression with a bounded number of leaves
		int depth = 0, leaves = 0, maxDepth = 0;
		for (int k = 0; k < c.node_count; k++) {
			const rt_csg_node& n = s->csg_nodes[c.node_first + k];
			if (n.kind == RT_CSG_UNION || n.kind == RT_CSG_INTERSECTION || n.kind == RT_CSG_DIFFERENCE) { RT_CHECK(depth >= 2, "csg program underflow"); depth--; }
			else if (n.kind == RT_CSG_SPHERE || n.kind == RT_CSG_BOX) { depth++; leaves++; }
			else if (n.kind == RT_CSG_MESH) { RT_CHECK(n.mesh >= 0 && n.mesh < s->n_meshes, "csg mesh index"); depth++; leaves++; }
			else RT_CHECK(false, "csg node kind");
			if (depth > maxDepth) maxDepth = depth;
		}
		RT_CHECK(depth == 1, "csg program does not reduce to one list");
		if (leaves > RT_MAX_CSG_LEAVES) { set_error("csg object has more leaves than RT_MAX_CSG_LEAVES"); return RT_ELIMIT; }
	}
	for (int i = 0; i < s->n_lights; i++) {
		const rt_light& L = s->lights[i];
		RT_CHECK(L.type >= RT_LIGHT_AMBIENT && L.type <= RT_LIGHT_CIRC, "light type");
		RT_CHECK(L.type < RT_LIGHT_RECT || (L.samples >= 0 && L.samples <= (1 << 20)), "area light sample count");
	}
	RT_CHECK((s->root_child < 0 && s->root_count == 0) || childOk(s, s->root_child, s->root_count), "root reference");
	for (int i = 0; i < s->n_bvh_nodes; i++)
		for (int k = 0; k < 2; k++) {
			RT_CHECK(childOk(s, s->bvh_nodes[i].child[k], s->bvh_nodes[i].count[k]), "bvh child reference");
			RT_CHECK(s->bvh_nodes[i].child[k] < 0 || s->bvh_nodes[i].child[k] > i, "bvh child must follow its parent (pre-order)");
		}
	for (int i = 0; i < s->n_leaf_prims; i++) {
		uint32_t ty = RT_PRIM_TYPE(s->leaf_prims[i]), ix = RT_PRIM_INDEX(s->leaf_prims[i]);
		int n = ty == RT_PRIM_SPHERE ? s->n_spheres : ty == RT_PRIM_TRI ? s->n_tris : ty == RT_PRIM_BOX ? s->n_boxes : ty == RT_PRIM_CSG ? s->n_csgs : -1;
		RT_CHECK(n >= 0 && (int64_t)ix < n, "leaf primitive reference");
	}
	return RT_OK;
}

}  // namespace rt

extern "C" {

int rt_camera_setup(double fov, int width, int height, const float eye[3], float rotX, float rotY, rt_camera* out) {
	if (!out || width <= 0 || height <= 0) { rt::set_error("rt_camera_setup: bad argument"); return RT_EINVAL; }
	const double PI = 3.14159265358979323846264338327950288;   // q1.cpp:7
	// setFacing(), q1.cpp:30-44, with d = 1 (q1.cpp:23)
	vec3 facing(-std::sin(rotY) * std::cos(rotX), std::sin(rotX), -std::cos(rotY) * std::cos(rotX));
	for (int i = 0; i < 3; i++)
		if (std::fabs(facing[i]) < 1e-5) facing[i] = 0;
	float aspect_ratio = (float)width / height;
	float h = 1.0f * (float)std::tan((PI * fov) / 180.0 / 2.0);
	float w = h * aspect_ratio;
	vec3 right = rt::normalize(rt::cross(vec3(-std::sin(rotY), 0, -std::cos(rotY)), vec3(0, 1, 0))) * w;
	vec3 up = rt::normalize(rt::cross(right, facing)) * h;
	for (int i = 0; i < 3; i++) {
		out->eye[i] = eye ? eye[i] : 0.0f;
		out->facing[i] = facing[i]; out->right[i] = right[i]; out->up[i] = up[i];
	}
	return RT_OK;
}

int rt_quantize_rgb8(const float* rgb, int width, int height, uint8_t* rgb8) {
	if (!rgb || !rgb8 || width <= 0 || height <= 0) { rt::set_error("rt_quantize_rgb8: bad argument"); return RT_EINVAL; }
	for (int y = 0; y < height; y++) {
		const float* src = rgb + (size_t)y * width * 3;                       // row 0 = bottom (q1.cpp:125-147)
		uint8_t* dst = rgb8 + (size_t)(height - 1 - y) * width * 3;           // row 0 = top
		for (int i = 0; i < width * 3; i++) {
			float c = src[i];
			if (!(c > 0.0f)) c = 0.0f;        // also catches NaN
			if (c > 1.0f) c = 1.0f;
			dst[i] = (uint8_t)(c * 255.0f + 0.5f);
		}
	}
	return RT_OK;
}

int rt_write_bmp(const char* path, const float* rgb, int width, int height) {
	if (!path || !rgb || width <= 0 || height <= 0) { rt::set_error("rt_write_bmp: bad argument"); return RT_EINVAL; }
	std::vector<uint8_t> px((size_t)width * height * 3);
	rt_quantize_rgb8(rgb, width, height, px.data());
	if (!rt::write_bmp(path, px.data(), width, height)) { rt::set_error(std::string("cannot write ") + path); return RT_EIO; }
	return RT_OK;
}

}  // extern "C"

// The reference narrates a picked ray on std::cout while it recurses (raytracer.cpp:307-327, objects.cpp:54-140): a hit,
// then the reflected ray's whole story, then -- after the lights -- the transmitted ray's.  The device produces the same
// facts level by level; here they are put back into that depth-first order, in the reference's words.  Numbers go through
// an ostream like the reference's (6 significant digits).
extern "C" size_t rt_pick_narration(const rt_pick_event* events, size_t n, char* buf, size_t cap) {
	std::map<uint32_t, std::vector<const rt_pick_event*>> byPath;
	for (size_t i = 0; i < n; i++) byPath[events[i].path].push_back(&events[i]);
	auto find = [&](uint32_t path, int kind) -> const rt_pick_event* {
		auto it = byPath.find(path);
		if (it == byPath.end()) return nullptr;
		for (const rt_pick_event* e : it->second) if (e->kind == kind) return e;
		return nullptr;
	};
	static const char* const typeNames[] = { "sphere", "triangle", "box", "csgobject", "plane" };   // Object::type, by RT_PRIM_*
	std::ostringstream out;
	auto point = [&](const float* p) { out << "{" << p[0] << ", " << p[1] << ", " << p[2] << "}"; };
	std::function<bool(uint32_t)> ray = [&](uint32_t path) -> bool {   // trace(): false = the caller reports the miss
		if (find(path, RT_PICK_MAXDEPTH)) { out << "Maximum number of reflections reached." << "\n"; return false; }
		const rt_pick_event* h = find(path, RT_PICK_HIT);
		if (!h) return false;
		out << "object " << (h->arg >= 0 && h->arg <= 4 ? typeNames[h->arg] : "object") << " hit at "; point(h->p); out << "\n";
		if (h->flags & 1) {   // objects.cpp:62-75
			out << "reflection:" << "\n";
			if (!ray(path * 2u)) out << "no additional objects hit" << "\n";
		}
		if (h->flags & 2) {   // objects.cpp:81-104
			out << "transmission:" << "\n";
			std::vector<const rt_pick_event*> inner;
			for (const rt_pick_event* e : byPath[path]) if (e->kind == RT_PICK_INTERIOR) inner.push_back(e);
			std::sort(inner.begin(), inner.end(), [](const rt_pick_event* a, const rt_pick_event* b) { return a->arg < b->arg; });
			for (const rt_pick_event* e : inner) { out << "interior reflection at "; point(e->p); out << "\n"; }
			if (const rt_pick_event* x = find(path, RT_PICK_EXIT)) {
				out << "exit object at "; point(x->p); out << "\n";
				if (!ray(path * 2u + 1u)) out << "no additional objects hit" << "\n";
			} else out << "ray lost due to too many internal reflections" << "\n";
		}
		return true;
	};
	ray(1u);
	const std::string text = out.str();
	if (buf && cap > 0) {
		const size_t m = std::min(text.size(), cap - 1);
		memcpy(buf, text.data(), m);
		buf[m] = 0;
	}
	return text.size();
}

