// raytracer_api.cpp -- the reference's Raytracer API (raytracer.h:12-18, raytracer.cpp:24-32,111-133,307-329) on top
// of the C-ABI: a process-global scene, choose_scene() with the reference's messages and exit(EXIT_FAILURE) on a
// missing file, and single-ray trace()/shadowRay() served by the GPU (batches of one; trace(..., pick = true) narrates like the reference does).
// No CPU path is involved.
#include "rt_capi.h"
#include "host_internal.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <iostream>
#include <string>
#include <vector>

// raytracer.cpp:24-25: the two globals callers read (include/raytracer.h declares them `extern double fov; extern colour3
// background_colour;` as raytracer.h:12-13 does; colour3 = glm::vec3 or rt_vec3, both three packed floats).  Namespace-scope
// variables are not mangled, so these are the symbols `fov` and `background_colour`.
struct rtapi_colour3 { float v[3]; };
__attribute__((visibility("default"))) double fov = 60;
__attribute__((visibility("default"))) rtapi_colour3 background_colour = { { 0, 0, 0 } };
static_assert(sizeof(rtapi_colour3) == 12 && alignof(rtapi_colour3) == 4, "layout of glm::vec3");

namespace {
const char* PATH = "scenes/";   // raytracer.cpp:22
rt_flat_scene* g_flat = nullptr;
rt_scene* g_scene = nullptr;

}  // namespace

extern "C" {

double* rtapi_fov(void) { return &fov; }
float* rtapi_background_colour(void) { return background_colour.v; }
rt_scene* rtapi_scene(void) { return g_scene; }
const rt_flat_scene* rtapi_flat_scene(void) { return g_flat; }

void rtapi_choose_scene(const char* fn) {
	if (fn == nullptr) {
		std::cout << "Using default input file " << PATH << "c.json\n";
		fn = "c";
	}
	std::cout << "Loading scene " << fn << std::endl;
	std::string fname = PATH + std::string(fn) + ".json";
	rt_flat_scene* flat = nullptr;
	int rc = rt_load_scene_json(fname.c_str(), PATH, &flat);
	if (rc == RT_EIO) {
		std::cout << "Unable to open scene file " << fname << std::endl;
		exit(EXIT_FAILURE);
	}
	if (rc != RT_OK) {   // the reference dies on an uncaught json exception here
		std::cerr << rt_last_error() << std::endl;
		abort();
	}
	if (g_scene) rt_scene_free(g_scene);
	if (g_flat) rt_flat_scene_free(g_flat);
	g_flat = flat; g_scene = nullptr;
	// raytracer.cpp:134-143: both keys are optional, and each one present is announced (glm::to_string(vec3) is "vec3(%f, %f, %f)")
	const rt::FlatSceneStore* store = rt::FlatSceneStore::from_view(flat);
	if (store->hasCameraField) {
		fov = flat->fov;
		std::cout << "Setting fov to " << fov << " degrees.\n";
	}
	if (store->hasCameraBackground) {
		for (int i = 0; i < 3; i++) background_colour.v[i] = flat->background[i];
		char text[96];
		snprintf(text, sizeof text, "vec3(%f, %f, %f)", background_colour.v[0], background_colour.v[1], background_colour.v[2]);
		std::cout << "Setting background colour to " << text << std::endl;
	}
	const char* dev = getenv("RT_DEVICE");
	rc = rt_scene_upload(flat, dev ? atoi(dev) : 0, &g_scene);
	if (rc != RT_OK) {
		std::cerr << "raytracer_amd: " << rt_last_error() << std::endl;
		exit(EXIT_FAILURE);
	}
}

int rtapi_trace(const float* e, const float* s, float* colour, int pick, int reflectionCount) {
	if (!g_scene) { std::cerr << "raytracer_amd: trace() before choose_scene()" << std::endl; abort(); }
	if (reflectionCount > RT_MAX_REFLECTIONS) {   // raytracer.cpp:308-313
		if (pick) std::cout << "Maximum number of reflections reached." << std::endl;
		colour[0] = colour[1] = colour[2] = 0;
		return 0;
	}
	const int depth = RT_MAX_REFLECTIONS - (reflectionCount > 0 ? reflectionCount : 0);
	uint8_t hit = 0; float rgb[3] = { 0, 0, 0 };
	int rc;
	if (pick) {   // the narration of raytracer.cpp:323-324 and objects.cpp:54-140, from the events the kernels leave
		std::vector<rt_pick_event> ev(1u << 16);
		size_t n = 0;
		rc = rt_trace_pick(g_scene, e, s, depth, 1, &hit, rgb, ev.data(), ev.size(), &n);
		if (rc == RT_OK) {
			n = std::min(n, ev.size());
			std::string text(rt_pick_narration(ev.data(), n, nullptr, 0) + 1, '\0');
			rt_pick_narration(ev.data(), n, &text[0], text.size());
			std::cout << text.c_str() << std::flush;
		}
	} else rc = rt_trace_batch(g_scene, e, s, 1, depth, 1, 0, &hit, nullptr, nullptr, rgb, nullptr);
	if (rc != RT_OK) { std::cerr << "raytracer_amd: " << rt_last_error() << std::endl; abort(); }
	if (!hit) return 0;   // colour untouched on a miss, like the reference
	colour[0] = rgb[0]; colour[1] = rgb[1]; colour[2] = rgb[2];
	return 1;
}

int rtapi_shadow_ray(const float* point, const float* lightPos, float* shadow) {
	if (!g_scene) { std::cerr << "raytracer_amd: shadowRay() before choose_scene()" << std::endl; abort(); }
	uint8_t lit = 0; float sh[3];
	int rc = rt_shadow_batch(g_scene, point, lightPos, 1, 0, &lit, sh, nullptr);
	if (rc != RT_OK) { std::cerr << "raytracer_amd: " << rt_last_error() << std::endl; abort(); }
	// the reference multiplies the caller's running attenuation in place (bvh.cpp:152)
	for (int i = 0; i < 3; i++) shadow[i] *= sh[i];
	return lit;
}

}  // extern "C"
