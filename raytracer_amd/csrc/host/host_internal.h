// host_internal.h -- owning storage behind an rt_flat_scene view, shared by the host translation units.
#pragma once
#include "rt_scene.h"
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace rt {

// The arrays that hold a record per triangle: resize() leaves new elements uninitialised (plain structs, every one is written
// right after), so that growing one by 5 M triangles does not first walk 320 MB on a single thread -- the loader fills such a
// range on several threads and each of them is then the first to touch its part.
template <class T> struct DefaultInitAllocator : std::allocator<T> {
	template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
	DefaultInitAllocator() = default;
	template <class U> DefaultInitAllocator(const DefaultInitAllocator<U>&) {}
	template <class U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
	template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using PodVec = std::vector<T, DefaultInitAllocator<T>>;

struct FlatSceneStore {
	rt_flat_scene view;   // must stay the first member: rt_flat_scene* handed out == FlatSceneStore*
	std::vector<rt_plane> planes;
	std::vector<rt_sphere> spheres;
	PodVec<rt_tri> tris;
	std::vector<rt_box> boxes;
	std::vector<rt_csg> csgs;
	std::vector<rt_csg_node> csg_nodes;
	std::vector<rt_mesh> meshes;
	PodVec<rt_tri> mesh_tris;
	PodVec<rt_uv> uvs;
	std::vector<rt_material> materials;
	std::vector<rt_light> lights;
	std::vector<rt_texture> textures;
	std::vector<uint8_t> texels;
	std::vector<rt_bvh_node> bvh_nodes;
	std::vector<uint32_t> leaf_prims;
	// which optional camera keys the scene file carried (raytracer.cpp:136-143 prints a line for each one present); not part of
	// the flat form on disk: a scene read from an .rtscene file reports neither
	bool hasCameraField = false, hasCameraBackground = false;
	FlatSceneStore() : view() {}
	void link();   // points the view at the vectors
	static FlatSceneStore* from_view(rt_flat_scene* v) { return reinterpret_cast<FlatSceneStore*>(v); }
	static const FlatSceneStore* from_view(const rt_flat_scene* v) { return reinterpret_cast<const FlatSceneStore*>(v); }
};

extern thread_local std::string g_lastError;
void set_error(const std::string& s);

// Checks every index a kernel will follow (child references, leaf primitive references, material / mesh /
// texture / uv indices, csg programs, texel ranges).  A scene that fails must never reach the GPU: an
// out-of-range access there faults the device.  Returns RT_OK or RT_EINVAL/RT_ELIMIT with the error set.
int validate_flat_scene(const rt_flat_scene* s);

}  // namespace rt
