// view_cli.cpp -- the reference's interactive session, scripted: the keys and mouse clicks a user of the GLUT window would make come from
// the command line (or stdin), every finished frame is written as a BMP, and what the reference prints on std::cout is printed here.
//   rt_view <scene-name|path.json|path.rtscene> [-w W] [-h H] [-gpu N] [-o prefix] [-progressive] <script>
// script: characters are keys (w s a d q e r f t g l p, space); '.' = display() a frame (-> <prefix>NNN.bmp); "c<x>,<y>;" = a left click at
// window position x, y (y from the top).  A frame is also displayed at the end if a key has changed the view since the last one.
// -progressive additionally writes, for the first frame, the sixteen partial images the reference shows while it fills the window
// (every 16th scanline per sweep, q1.cpp:124).  Example:  rt_view c -w 640 -h 480 ".wwq.l.c320,240;"
#include "rt_capi.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool endsWith(const std::string& s, const char* suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

int main(int argc, char** argv) {
	std::string scene = "c", prefix = "frame", script;
	int W = 512, H = 512, gpu = 0;
	bool haveScene = false, progressive = false;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		if (a == "-w" && i + 1 < argc) W = atoi(argv[++i]);
		else if (a == "-h" && i + 1 < argc) H = atoi(argv[++i]);
		else if (a == "-gpu" && i + 1 < argc) gpu = atoi(argv[++i]);
		else if (a == "-o" && i + 1 < argc) prefix = argv[++i];
		else if (a == "-progressive") progressive = true;
		else if (!haveScene) { scene = a; haveScene = true; }
		else script += a;
	}
	if (script.empty()) { char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, stdin)) > 0) script.append(buf, n); }
	rt_flat_scene* flat = nullptr;
	int rc;
	if (endsWith(scene, ".rtscene")) rc = rt_flat_scene_load(scene.c_str(), &flat);
	else if (endsWith(scene, ".json")) rc = rt_load_scene_json(scene.c_str(), nullptr, &flat);
	else rc = rt_load_scene_json(("scenes/" + scene + ".json").c_str(), "scenes/", &flat);
	if (rc != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
	rt_scene* sc = nullptr;
	if ((rc = rt_scene_upload(flat, gpu, &sc)) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
	rt_view* view = nullptr;
	if ((rc = rt_view_create(sc, flat->fov, W, H, &view)) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
	std::vector<float> rgb((size_t)W * H * 3);
	int frames = 0;
	auto display = [&]() -> int {
		rt_stats st;
		if (int r = rt_view_render_host(view, rgb.data(), &st)) return r;
		char name[512];
		if (progressive && frames == 0) {   // the window as the reference shows it while display() fills it: 16 sweeps of every 16th scanline
			std::vector<float> part((size_t)W * H * 3, 0.0f);
			const int perSweep = (H + 15) / 16;
			for (int pass = 0; pass < H; pass++) {
				const int y = rt_view_scanline(H, pass);
				if (y >= 0 && y < H) memcpy(&part[(size_t)y * W * 3], &rgb[(size_t)y * W * 3], (size_t)W * 3 * sizeof(float));
				if ((pass + 1) % perSweep == 0 || pass == H - 1) {
					snprintf(name, sizeof name, "%s%03d_sweep%02d.bmp", prefix.c_str(), frames, pass / perSweep);
					if (int r = rt_write_bmp(name, part.data(), W, H)) return r;
				}
			}
		}
		snprintf(name, sizeof name, "%s%03d.bmp", prefix.c_str(), frames);
		if (int r = rt_write_bmp(name, rgb.data(), W, H)) return r;
		printf("[frame %d] %s  %llu rays, %.3f ms on the GPU\n", frames, name, (unsigned long long)(st.rays_nearest + st.rays_shadow), st.gpu_ms);
		frames++;
		return RT_OK;
	};
	bool dirty = true;
	char text[1 << 16];
	for (size_t i = 0; i < script.size(); i++) {
		const char ch = script[i];
		if (ch == '\n' || ch == '\r') continue;
		if (ch == '.') { if ((rc = display())) break; dirty = false; continue; }
		if (ch == 'c' && i + 1 < script.size() && (script[i + 1] >= '0' && script[i + 1] <= '9')) {
			int x = 0, y = 0, used = 0;
			if (sscanf(script.c_str() + i + 1, "%d,%d;%n", &x, &y, &used) == 2 && used > 0) {
				if (dirty) { if ((rc = display())) break; dirty = false; }   // a click is answered for the frame on screen
				size_t n = 0;
				if ((rc = rt_view_click(view, x, y, text, sizeof text, &n))) break;
				fputs(text, stdout);
				i += (size_t)used;
				continue;
			}
		}
		size_t n = 0;
		const int r = rt_view_key(view, (unsigned char)ch, text, sizeof text, &n);
		if (n) fputs(text, stdout);
		if (r == -1) break;
		if (r == 1) dirty = true;
	}
	if (rc == RT_OK && dirty) rc = display();
	if (rc != RT_OK) fprintf(stderr, "%s\n", rt_last_error());
	rt_view_free(view);
	rt_scene_free(sc);
	rt_flat_scene_free(flat);
	return rc == RT_OK ? 0 : 1;
}
