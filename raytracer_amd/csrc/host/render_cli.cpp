// render_cli.cpp -- headless stand-in for the reference's GLUT front end (main.cpp:107-129, q1.cpp:107-175):
// loads a scene through the Raytracer API, renders one frame on the GPU and writes it as a BMP.
//   rt_render <scene-name|path.json|path.rtscene> [-o out.bmp] [-w W] [-h H] [-aa N] [-depth D] [-eye x y z] [-rot rx ry] [-gpu N]
// A bare scene name is resolved like the reference does: scenes/<name>.json relative to the working directory.
#include "rt_capi.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool endsWith(const std::string& s, const char* suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

int main(int argc, char** argv) {
	std::string scene = "c", out = "out.bmp";
	int W = 512, H = 512, aa = 1, depth = RT_MAX_REFLECTIONS, gpu = 0;   // 512x512 is the reference's window (main.cpp:112)
	float eye[3] = { 0, 0, 0 }, rotX = 0, rotY = 0;
	bool haveScene = false;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto need = [&](int k) { if (i + k >= argc) { fprintf(stderr, "missing value after %s\n", a.c_str()); exit(2); } };
		if (a == "-o") { need(1); out = argv[++i]; }
		else if (a == "-w") { need(1); W = atoi(argv[++i]); }
		else if (a == "-h") { need(1); H = atoi(argv[++i]); }
		else if (a == "-aa") { need(1); aa = atoi(argv[++i]); }
		else if (a == "-depth") { need(1); depth = atoi(argv[++i]); }
		else if (a == "-gpu") { need(1); gpu = atoi(argv[++i]); }
		else if (a == "-eye") { need(3); for (int k = 0; k < 3; k++) eye[k] = (float)atof(argv[++i]); }
		else if (a == "-rot") { need(2); rotX = (float)atof(argv[++i]); rotY = (float)atof(argv[++i]); }
		else if (!haveScene) { scene = a; haveScene = true; }
		else { fprintf(stderr, "unexpected argument %s\n", a.c_str()); return 2; }
	}
	rt_flat_scene* flat = nullptr;
	int rc;
	if (endsWith(scene, ".rtscene")) rc = rt_flat_scene_load(scene.c_str(), &flat);
	else if (endsWith(scene, ".json")) rc = rt_load_scene_json(scene.c_str(), nullptr, &flat);
	else rc = rt_load_scene_json(("scenes/" + scene + ".json").c_str(), "scenes/", &flat);
	if (rc != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
	rt_scene* sc = nullptr;
	if ((rc = rt_scene_upload(flat, gpu, &sc)) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
	rt_camera cam;
	rt_camera_setup(flat->fov, W, H, eye, rotX, rotY, &cam);
	std::vector<float> rgb((size_t)W * H * 3);
	rt_stats st;
	if ((rc = rt_render_host(sc, &cam, W, H, aa, aa, depth, 1, 0, rgb.data(), &st)) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
	if ((rc = rt_write_bmp(out.c_str(), rgb.data(), W, H)) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
	const double rays = (double)st.rays_nearest + (double)st.rays_shadow;
	printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"aa\": %d, \"rays_nearest\": %llu, \"rays_shadow\": %llu, \"gpu_ms\": %.3f, \"mrays_per_s\": %.1f, \"out\": \"%s\"}\n",
	       scene.c_str(), W, H, aa, (unsigned long long)st.rays_nearest, (unsigned long long)st.rays_shadow, st.gpu_ms, rays / (st.gpu_ms * 1e3), out.c_str());
	rt_scene_free(sc);
	rt_flat_scene_free(flat);
	return 0;
}
