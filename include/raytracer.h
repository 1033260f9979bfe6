/* raytracer.h -- source-compatible stand-in for the reference's /root/reference/src/raytracer.h:12-18.
 *
 * The reference's frame driver (q1.cpp) includes "raytracer.h" and calls choose_scene / trace / shadowRay and reads
 * the globals fov and background_colour.  Including THIS header instead (and linking libraytracer_amd.so) routes
 * those calls to the MI355X path: choose_scene additionally flattens and uploads the scene, trace and shadowRay
 * become one-element batches on the GPU.  For whole frames use rt_render (rt_capi.h) -- one call per frame instead
 * of one per pixel; see INTEGRATION.md.
 *
 * point3/colour3 are glm::vec3 when RT_USE_GLM is defined (as in the reference), otherwise a POD of three floats
 * with the same layout.
 */
#ifndef RAYTRACER_H
#define RAYTRACER_H

#include "rt_capi.h"

#define MAX_T 10000.
#define MAX_REFLECTIONS 16

#ifdef RT_USE_GLM
#include <glm/glm.hpp>
typedef glm::vec3 point3;
typedef glm::vec3 colour3;
#else
struct rt_vec3 {
	float x, y, z;
	rt_vec3() : x(0), y(0), z(0) {}
	rt_vec3(float a, float b, float c) : x(a), y(b), z(c) {}
	float& operator[](int i) { return (&x)[i]; }
	const float& operator[](int i) const { return (&x)[i]; }
};
typedef rt_vec3 point3;
typedef rt_vec3 colour3;
#endif

/* The two globals of raytracer.h:12-13, real objects exported by libraytracer_amd.so (csrc/host/raytracer_api.cpp defines
 * `fov` as a double and `background_colour` as 12 bytes of three floats: the layout of glm::vec3 and of rt_vec3).  No macros:
 * a caller may have members and locals called fov. */
extern double fov;
extern colour3 background_colour;

extern "C" {
/* exported by libraytracer_amd.so (csrc/host/raytracer_api.cpp) */
void rtapi_choose_scene(const char* fn);
int rtapi_trace(const float* e, const float* s, float* colour, int pick, int reflectionCount);
int rtapi_shadow_ray(const float* point, const float* lightPos, float* shadow);
double* rtapi_fov(void);                  /* &fov, for bindings that cannot name a data symbol */
float* rtapi_background_colour(void);     /* &background_colour */
rt_scene* rtapi_scene(void);
const rt_flat_scene* rtapi_flat_scene(void);
}

inline void choose_scene(char const* fn) { rtapi_choose_scene(fn); }
inline bool trace(const point3& e, const point3& s, colour3& colour, bool pick, int reflectionCount = 0) {
	return rtapi_trace(&e[0], &s[0], &colour[0], pick ? 1 : 0, reflectionCount) != 0;
}
inline bool shadowRay(const point3& point, const point3& lightPos, point3& shadow) {
	return rtapi_shadow_ray(&point[0], &lightPos[0], &shadow[0]) != 0;
}

#endif
