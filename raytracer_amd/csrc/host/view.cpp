// view.cpp -- the reference's interactive front end (q1.cpp:23-29, 107-288) without its window: the camera that the WASD / QE / RF / TG
// keys move, the `l` key's 4x supersampling, the `p` key's print-out, the mouse pick with its narration, reshape, and the order in
// which display() fills the window (every 16th scanline, 16 passes).  The reference draws one scanline per GLUT timer tick and calls
// trace() once per pixel; here display() is ONE rt_render call on the GPU, and the scanline order is exposed for a front end that
// wants to reveal the frame the way the reference does.  No GL: there is no display on a GPU box.
#include "rt_capi.h"
#include "rt_vec.h"
#include "host_internal.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

using rt::vec3;

struct rt_view {
	rt_scene* scene = nullptr;
	double fov = 60;
	int width = 512, height = 512;                 // main.cpp:112
	vec3 eye;                                      // q1.cpp:22: the origin
	float rotationX = 0, rotationY = 0;            // q1.cpp:26
	bool antialias = false;                        // q1.cpp:47
	bool dirty = true;                             // drawing_y == 0: the next display() starts a new frame
	rt_camera basis{};                             // facing / camera_right / camera_up as the LAST display() left them (q1.cpp:111-120)
};

namespace {
const double PI = 3.14159265358979323846264338327950288;   // q1.cpp:7
const float move_step = 0.5f;                                // q1.cpp:27
const float rotate_step = (float)(PI / 8);                   // q1.cpp:28

void setFacing(rt_view* v) {
	const float eye[3] = { v->eye.x, v->eye.y, v->eye.z };
	rt_camera_setup(v->fov, v->width, v->height, eye, v->rotationX, v->rotationY, &v->basis);
}
size_t emit(const std::string& s, char* text, size_t cap) {
	if (text && cap > 0) { const size_t n = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(text, s.data(), n); text[n] = 0; }
	return s.size();
}
}  // namespace

extern "C" {

int rt_view_create(rt_scene* scene, double fov, int width, int height, rt_view** out) {
	if (!out || width <= 0 || height <= 0) { rt::set_error("rt_view_create: bad argument"); return RT_EINVAL; }
	rt_view* v = new rt_view();
	v->scene = scene; v->fov = fov; v->width = width; v->height = height;
	setFacing(v);   // (the reference's globals are zero until the first display(); a key pressed before that moves along a zero vector --
	                //  here the basis of the initial pose is there from the start)
	*out = v;
	return RT_OK;
}
void rt_view_free(rt_view* v) { delete v; }

int rt_view_reshape(rt_view* v, int width, int height) {   // q1.cpp:278-288
	if (!v || width <= 0 || height <= 0) { rt::set_error("rt_view_reshape: bad argument"); return RT_EINVAL; }
	v->width = width; v->height = height; v->dirty = true;
	return RT_OK;
}

// keyboard(), q1.cpp:179-247.  Returns 1 if the key starts a new frame (drawing_y = 0), 0 if not, -1 for Escape (the reference exits),
// -2 on a bad argument (a NULL view); *text_len receives the length of what the reference prints on std::cout for this key (`p`, `l`), written to text.
int rt_view_key(rt_view* v, unsigned char key, char* text, size_t cap, size_t* text_len) {
	if (text && cap > 0) text[0] = 0;
	if (text_len) *text_len = 0;
	if (!v) { rt::set_error("rt_view_key: bad argument"); return -2; }
	const vec3 facing(v->basis.facing[0], v->basis.facing[1], v->basis.facing[2]);
	const vec3 right(v->basis.right[0], v->basis.right[1], v->basis.right[2]);
	bool redraw = false;
	std::ostringstream os;
	switch (key) {
	case 033: return -1;
	case ' ': break;                                   // drawing_y = 1: draw the same frame again
	case 'w': v->eye = v->eye + facing * move_step; redraw = true; break;
	case 's': v->eye = v->eye - facing * move_step; redraw = true; break;
	case 'a': v->eye = v->eye - rt::normalize(right) * move_step; redraw = true; break;
	case 'd': v->eye = v->eye + rt::normalize(right) * move_step; redraw = true; break;
	case 'q': v->rotationY += rotate_step; redraw = true; break;
	case 'e': v->rotationY -= rotate_step; redraw = true; break;
	case 'r': if (v->rotationX < PI / 2) { v->rotationX += rotate_step; redraw = true; } break;
	case 'f': if (v->rotationX > -PI / 2) { v->rotationX -= rotate_step; redraw = true; } break;
	case 't': v->eye = v->eye + vec3(0, move_step, 0); redraw = true; break;
	case 'g': v->eye = v->eye + vec3(0, -move_step, 0); redraw = true; break;
	case 'p':
		os << "camera: (" << v->eye.x << ", " << v->eye.y << ", " << v->eye.z << ")" << std::endl;
		os << "facing: <" << facing.x << ", " << facing.y << ", " << facing.z << ">" << std::endl << std::endl;
		break;
	case 'l':
		v->antialias = !v->antialias;
		os << (v->antialias ? "Anti-aliasing ON" : "Anti-aliasing OFF") << std::endl;
		redraw = true;
		break;
	default: break;
	}
	if (redraw) v->dirty = true;
	const size_t n = emit(os.str(), text, cap);
	if (text_len) *text_len = n;
	return redraw ? 1 : 0;
}

// The pose the next display() will use: setFacing() with the current eye and rotation (q1.cpp:30-44), and the sample grid (1, or 2 = the
// four s_aa samples of q1.cpp:55-64).  Does not change what the keys move along.
int rt_view_camera(const rt_view* v, rt_camera* camera, int* aa) {
	if (!v || !camera) { rt::set_error("rt_view_camera: bad argument"); return RT_EINVAL; }
	const float eye[3] = { v->eye.x, v->eye.y, v->eye.z };
	if (aa) *aa = v->antialias ? 2 : 1;
	return rt_camera_setup(v->fov, v->width, v->height, eye, v->rotationX, v->rotationY, camera);
}
int rt_view_state(const rt_view* v, float eye[3], float* rotation_x, float* rotation_y, int* width, int* height, int* needs_redraw) {
	if (!v) { rt::set_error("rt_view_state: bad argument"); return RT_EINVAL; }
	if (eye) { eye[0] = v->eye.x; eye[1] = v->eye.y; eye[2] = v->eye.z; }
	if (rotation_x) *rotation_x = v->rotationX;
	if (rotation_y) *rotation_y = v->rotationY;
	if (width) *width = v->width;
	if (height) *height = v->height;
	if (needs_redraw) *needs_redraw = v->dirty ? 1 : 0;
	return RT_OK;
}

// display(), q1.cpp:107-175, as one frame: setFacing(), then every pixel through trace() -- one rt_render on the GPU.
int rt_view_render(rt_view* v, float* rgb_dev, void* stream, rt_stats* stats) {
	if (!v || !v->scene || !rgb_dev) { rt::set_error("rt_view_render: bad argument"); return RT_EINVAL; }
	setFacing(v);
	const int aa = v->antialias ? 2 : 1;
	const int rc = rt_render(v->scene, &v->basis, v->width, v->height, aa, aa, RT_MAX_REFLECTIONS, 0, 1, 1u, 0u, rgb_dev, stream, stats);
	if (rc == RT_OK) v->dirty = false;
	return rc;
}

// The same into a HOST buffer (rt_render_host underneath: a temporary device frame and one copy).
int rt_view_render_host(rt_view* v, float* rgb_host, rt_stats* stats) {
	if (!v || !v->scene || !rgb_host) { rt::set_error("rt_view_render_host: bad argument"); return RT_EINVAL; }
	setFacing(v);
	const int aa = v->antialias ? 2 : 1;
	const int rc = rt_render_host(v->scene, &v->basis, v->width, v->height, aa, aa, RT_MAX_REFLECTIONS, 1u, 0u, rgb_host, stats);
	if (rc == RT_OK) v->dirty = false;
	return rc;
}

// The scanline display() draws in its pass number `pass` (0 .. height - 1): every 16th row, 16 sweeps (q1.cpp:124 with drawing_y = pass + 1).
int rt_view_scanline(int height, int pass) {
	if (height <= 0 || pass < 0) return -1;
	return ((pass * 16) % height) + (((pass * 16) / height) * 7) % 16;
}

// mouse(GLUT_LEFT_BUTTON, GLUT_DOWN, x, y), q1.cpp:251-269: y counts from the top of the window; the picked ray is narrated by trace(...,
// pick = true) and the HIT / MISS lines follow.  The text the reference prints on std::cout is written to `text`.
int rt_view_click(rt_view* v, int x, int y_from_top, char* text, size_t cap, size_t* text_len) {
	if (text && cap > 0) text[0] = 0;
	if (text_len) *text_len = 0;
	if (!v || !v->scene) { rt::set_error("rt_view_click: bad argument"); return RT_EINVAL; }
	const int y = v->height - y_from_top - 1;
	// s(x, y), q1.cpp:51-53, reads the globals as they are NOW: `eye` is moved by the w/s/a/d/t/g keys at once, while facing / camera_right /
	// camera_up are only rewritten by the next display() (setFacing, q1.cpp:111-120).  A click between a move key and the next frame
	// therefore starts at the moved eye with the basis of the frame still on screen -- and so does trace(eye, uvw, ...), q1.cpp:259.
	const rt_camera& c = v->basis;
	const vec3 eye = v->eye, facing(c.facing[0], c.facing[1], c.facing[2]), right(c.right[0], c.right[1], c.right[2]), up(c.up[0], c.up[1], c.up[2]);
	const vec3 uvw = eye + facing + right * (2 * ((x + 0.5f) / v->width - 0.5f)) + up * (2 * ((y + 0.5f) / v->height - 0.5f));
	std::vector<rt_pick_event> ev(1u << 16);
	size_t n = 0;
	uint8_t hit = 0;
	float rgb[3] = { 0, 0, 0 };
	const float e[3] = { eye.x, eye.y, eye.z }, s[3] = { uvw.x, uvw.y, uvw.z };
	if (int rc = rt_trace_pick(v->scene, e, s, RT_MAX_REFLECTIONS, 1u, &hit, rgb, ev.data(), ev.size(), &n)) return rc;
	n = n < ev.size() ? n : ev.size();
	std::string narration(rt_pick_narration(ev.data(), n, nullptr, 0) + 1, '\0');
	rt_pick_narration(ev.data(), n, &narration[0], narration.size());
	std::ostringstream os;
	os << std::endl << narration.c_str();
	if (hit) {
		os << "HIT @ ( " << uvw.x << "," << uvw.y << "," << uvw.z << " )\n";
		os << "      colour = ( " << rgb[0] << "," << rgb[1] << "," << rgb[2] << " )\n";
	} else os << "MISS @ ( " << uvw.x << "," << uvw.y << "," << uvw.z << " )\n";
	os << std::endl;
	const size_t len = emit(os.str(), text, cap);
	if (text_len) *text_len = len;
	return RT_OK;
}

}  // extern "C"
