/* rt_scene.h -- the flat scene IR that crosses the C-ABI (plain structs, pointers and counts; C and C++).
 *
 * The reference keeps its scene as a graph of heap objects with virtual dispatch
 * (std::vector<Object*> Objects, std::vector<Light*> Lights, BVH* bvh -- raytracer.cpp:29-32).  The MI355X path
 * wants the same information as dense arrays that can be copied to HBM once and indexed by lane:
 *
 *   planes[]      the side list BVH::planes (bvh.cpp:13-16)             32 B records
 *   spheres[]     Sphere / BumpSphere (objects.h:80, bump.h:8)          32 B
 *   tris[]        Triangle / TextureTriangle exploded from meshes       64 B, in BVH leaf (DFS) order
 *   boxes[]       Box (objects.h:125)                                   32 B
 *   csgs[] + csg_nodes[]   csgObject + its csg_node tree in post-order  (csg.h:15-37)
 *   meshes[] + mesh_tris[] Mesh::triangles in scene order, for Mesh::rayhit's linear scan (objects.cpp:286-304)
 *   bvh_nodes[]   the reference's median-split tree (bvh.cpp:37-70) flattened into 64 B two-child nodes
 *   leaf_prims[]  BVH_node::objects of every leaf, concatenated in left-to-right DFS order
 *
 * Object ids (the `id` field everywhere) number the reference's objects the way its BVH sees them:
 * planes in scene order first, then the BVH object list = scene order with "mesh"-typed objects replaced by
 * their triangles (bvh.cpp:10-28).
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RT_PRIM_SPHERE = 0, RT_PRIM_TRI = 1, RT_PRIM_BOX = 2, RT_PRIM_CSG = 3, RT_PRIM_PLANE = 4, RT_PRIM_NONE = 15 };
#define RT_PRIM_TYPE(ref)  (((uint32_t)(ref) >> 28) & 7u)
#define RT_PRIM_INDEX(ref) ((uint32_t)(ref) & 0x0fffffffu)
#define RT_PRIM_REF(type, index) (((uint32_t)(type) << 28) | (uint32_t)(index))
#define RT_PRIM_LAST 0x80000000u   /* set on the last entry of every leaf's run in leaf_prims[] */
#define RT_NO_HIT 0xffffffffu

enum { RT_LIGHT_AMBIENT = 0, RT_LIGHT_DIRECTIONAL = 1, RT_LIGHT_POINT = 2, RT_LIGHT_SPOT = 3, RT_LIGHT_RECT = 4, RT_LIGHT_CIRC = 5 };
enum { RT_CSG_UNION = 0, RT_CSG_INTERSECTION = 1, RT_CSG_DIFFERENCE = 2, RT_CSG_SPHERE = 3, RT_CSG_BOX = 4, RT_CSG_MESH = 5 };

#define RT_MAX_T 10000.0f          /* raytracer.h:6, bvh.cpp:4 */
#define RT_MAX_REFLECTIONS 16      /* raytracer.h:7 */
#define RT_MAX_BVH_DEPTH 16        /* bvh.h:6 */
#define RT_MAX_CSG_LEAVES 16       /* capacity of the fixed-size interval lists on the device */

/* struct Material, objects.h:21-29 (20 floats so records stay 16 B aligned) */
typedef struct rt_material {
	float ambient[3], diffuse[3], specular[3];
	float shininess;
	float reflective[3], transmissive[3];
	float refraction;
	float pad[3];
} rt_material;

typedef struct rt_plane  { float point[3], normal[3]; int32_t id, mat; } rt_plane;
typedef struct rt_sphere { float center[3], radius; int32_t id, mat, bump_tex; float bump_depth; } rt_sphere; /* bump_tex < 0: plain Sphere */
typedef struct rt_tri    { float p0[3], p1[3], p2[3], n[3]; int32_t id, mat, mesh, uv; } rt_tri;          /* uv < 0: no TextureTriangle data */
typedef struct rt_box    { float b[6]; int32_t id, mat; } rt_box;                                            /* minX,maxX,minY,maxY,minZ,maxZ (objects.h:17) */
typedef struct rt_csg    { int32_t id, mat, node_first, node_count; float b[6]; int32_t pad[2]; } rt_csg;   /* b = csgObject::setBox result */
/* post-order (RPN) csg_node: leaves push an interval list, operations pop second then first (csg.cpp:115-116) */
typedef struct rt_csg_node { int32_t kind, mesh; float f[6]; } rt_csg_node;   /* sphere: f = cx,cy,cz,r ; box: f = b[6] ; mesh: index into meshes[] */
typedef struct rt_mesh   { int32_t tri_first, tri_count, tex, mat; } rt_mesh; /* tex < 0: plain Mesh */
typedef struct rt_uv     { float uv[6]; } rt_uv;                              /* TextureTriangle::uvCoords, texturemesh.h:20 */
typedef struct rt_texture { int32_t width, height; int64_t offset; } rt_texture; /* RGB8, texel (i,j) at offset + 3*(j*width+i); j = 0 is the TOP row (EasyBMP) */

typedef struct rt_light {
	int32_t type, samples;
	float colour[3];
	float position[3];
	float direction[3];      /* normalised at load, objects.cpp:395,440 */
	float normal[3];         /* area lights (not used after construction, kept for completeness) */
	float planeX[3], planeY[3];
	float width, height, radius, cutoff;
	double cos_cutoff;       /* cos(cutoff * M_PI / 180) in double, objects.cpp:451 */
} rt_light;

/* Two-child node: the boxes tested are the CHILDREN's BVH_node::boundingBox, so one 64 B read serves both
 * recursive calls of findRecursive/shadowRecursive (bvh.cpp:104-105,138-141).
 * child >= 0: index of an inner node.  child < 0: leaf, leaf_prims[~child .. ~child + count).  */
typedef struct rt_bvh_node { float box[2][6]; int32_t child[2]; int32_t count[2]; } rt_bvh_node;

typedef struct rt_flat_scene {
	double fov;                  /* raytracer.cpp:24,137 */
	float background[3];         /* raytracer.cpp:25,141 */
	int32_t n_objects;           /* number of ids */
	float root_box[6];           /* bvh->root->boundingBox */
	int32_t root_child, root_count; /* same encoding as rt_bvh_node.child/count; root_count == 0 && root_child < 0: empty tree */
	int32_t n_planes, n_spheres, n_tris, n_boxes, n_csgs, n_csg_nodes, n_meshes, n_mesh_tris, n_uvs,
	        n_materials, n_lights, n_textures, n_bvh_nodes, n_leaf_prims;
	int64_t n_texel_bytes;
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
	const rt_bvh_node* bvh_nodes;
	const uint32_t* leaf_prims;  /* RT_PRIM_REF(type, index into the typed array) | RT_PRIM_LAST on a leaf's last entry */
} rt_flat_scene;

/* Camera basis as q1.cpp:30-44 leaves it: right/up already scaled by w and h. */
typedef struct rt_camera { float eye[3], facing[3], right[3], up[3]; } rt_camera;

#ifdef __cplusplus
}
#endif
#endif
