// tests/shim_client.cpp -- a caller written against the reference's API (raytracer.h:12-18) the way q1.cpp uses it: globals
// fov / background_colour, choose_scene, trace per pixel (with and without pick), shadowRay.  Built against THIS repository's
// include/raytracer.h and linked with libraytracer_amd.so by the tests; test infrastructure, not part of the product.
//   shim_client <scene> <W> <H> [x y ...]    run with cwd = the directory that holds scenes/<scene>.json
#include "raytracer.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>

// a caller with names of its own called fov: a member, a local, and the flat scene's field (raytracer.h used to define fov as a macro)
struct Lens { double fov; };
static double widest(const Lens& a, const rt_flat_scene* flat) {
	double fov = a.fov;
	if (flat && flat->fov > fov) fov = flat->fov;
	return fov;
}

int main(int argc, char** argv) {
	if (argc < 4) return 2;
	choose_scene(argv[1]);
	if (widest(Lens{ 1.0 }, rtapi_flat_scene()) != ::fov || *rtapi_fov() != ::fov) return 3;
	if (&rtapi_background_colour()[0] != &background_colour[0]) return 4;
	const int W = atoi(argv[2]), H = atoi(argv[3]);
	colour3 bg = background_colour;
	printf("fov %.9g background %.9g %.9g %.9g\n", fov, bg[0], bg[1], bg[2]);
	// the camera of q1.cpp:30-53 at the origin, facing -z (rt_camera_setup is the library's restatement of it)
	rt_camera cam;
	const float origin[3] = { 0, 0, 0 };
	rt_camera_setup(fov, W, H, origin, 0.0f, 0.0f, &cam);
	for (int a = 4; a + 1 < argc; a += 2) {
		const int x = atoi(argv[a]), y = atoi(argv[a + 1]);
		const float fa = 2 * ((x + 0.5f) / W - 0.5f), fb = 2 * ((y + 0.5f) / H - 0.5f);
		point3 e(0, 0, 0), s;
		for (int c = 0; c < 3; c++) s[c] = ((cam.eye[c] + cam.facing[c]) + cam.right[c] * fa) + cam.up[c] * fb;
		std::cout << "== pixel " << x << " " << y << std::endl;
		colour3 colour(-1, -1, -1), again(-1, -1, -1);
		const bool hit = trace(e, s, colour, true);      // narrates on std::cout
		const bool hit2 = trace(e, s, again, false);
		std::cout << "== " << (hit ? "hit" : "miss") << std::endl;
		printf("colour %.9g %.9g %.9g same %d\n", colour[0], colour[1], colour[2],
		       hit == hit2 && colour[0] == again[0] && colour[1] == again[1] && colour[2] == again[2]);
		if (hit) {
			point3 light(0.5f, 6.0f, -1.0f), shadow(1, 1, 1);
			const bool lit = shadowRay(s, light, shadow);
			printf("shadowRay %d %.9g %.9g %.9g\n", lit ? 1 : 0, shadow[0], shadow[1], shadow[2]);
		}
	}
	colour3 c(7, 7, 7);
	const bool deep = trace(point3(0, 0, 0), point3(0, 0, -1), c, true, MAX_REFLECTIONS + 1);   // raytracer.cpp:308-313
	printf("beyond the depth limit: %d %.9g %.9g %.9g\n", deep ? 1 : 0, c[0], c[1], c[2]);
	return 0;
}
