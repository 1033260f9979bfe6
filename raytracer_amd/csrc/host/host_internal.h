// host_internal.h -- owning storage behind an rt_flat_scene view, shared by the host translation units.
#pragma once
#include "rt_scene.h"
#include <string>
#include <vector>

namespace rt {

struct FlatSceneStore {
	rt_flat_scene view;   // must stay the first member: rt_flat_scene* handed out == FlatSceneStore*
	std::vector<rt_plane> planes;
	std::vector<rt_sphere> spheres;
	std::vector<rt_tri> tris;
	std::vector<rt_box> boxes;
	std::vector<rt_csg> csgs;
	std::vector<rt_csg_node> csg_nodes;
	std::vector<rt_mesh> meshes;
	std::vector<rt_tri> mesh_tris;
	std::vector<rt_uv> uvs;
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
