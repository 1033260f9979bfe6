// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Drives the UNMODIFIED reference sources (compiled in place from /root/reference/src by oracle/Makefile
// into oracle/_ref/) and dumps what they compute, so that the CPU restatement in oracle/restate.cpp and
// the HIP path can be pinned against the real thing.  Runs only in the build container (cwd must be
// /root/reference/src because the reference opens "scenes/<name>.json" relative to cwd,
// raytracer.cpp:22,125); its outputs become the committed fixtures under tests/golden/.
//
// What it calls in the reference:
//   choose_scene()                       raytracer.cpp:117
//   trace()                              raytracer.cpp:307
//   BVH::findNearest / BVH::calcShadow   bvh.cpp:72,123   (public members, reached through `extern BVH* bvh`)
// The camera math below is this file's own restatement of q1.cpp:30-53 (q1.cpp itself needs OpenGL).
//
// BVH::findNearest and BVH::calcShadow are additionally intercepted with `ld --wrap` so that the secondary
// rays issued inside trace() (reflection, transmission, shadow) can be counted and sampled.
#include "raytracer.h"
#include "objects.h"
#include "bvh.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <vector>
#include <string>

extern std::vector<Object*> Objects;
extern std::vector<Light*> Lights;
extern BVH* bvh;

// ---- object ids: planes in scene order, then the BVH object list (scene order, meshes exploded) ------
// mirrors the partition in BVH::BVH, bvh.cpp:10-28
static std::map<Object*, int> g_ids;
static void build_ids() {
	int next = 0;
	for (size_t i = 0; i < Objects.size(); i++)
		if (Objects[i]->type == "plane") g_ids[Objects[i]] = next++;
	for (size_t i = 0; i < Objects.size(); i++) {
		Object* o = Objects[i];
		if (o->type == "plane") continue;
		if (o->type == "mesh") {
			Mesh* m = (Mesh*)o;
			for (size_t t = 0; t < m->triangles.size(); t++) g_ids[m->triangles[t]] = next++;
		} else g_ids[o] = next++;
	}
}

// ---- capture of the calls made inside trace() ----------------------------------------------------------
struct NearRec { float e[3], d[3]; int32_t id; float t; };
struct ShadRec { float p[3], l[3]; int32_t lit; float shadow[3]; };
static std::vector<NearRec> g_near;
static std::vector<ShadRec> g_shad;
static uint64_t g_nNear = 0, g_nShad = 0;       // all calls, including primary
static uint64_t g_nSecondary = 0;                // findNearest calls that are not the first of a trace()
static bool g_primaryPending = false;
static uint64_t g_strideNear = 0, g_strideShad = 0;   // 0 = capture off

extern "C" Object* __real__ZN3BVH11findNearestEN3glm5tvec3IfLNS0_9precisionE0EEES3_(BVH*, point3, point3);
extern "C" bool __real__ZN3BVH10calcShadowEN3glm5tvec3IfLNS0_9precisionE0EEES3_RS3_(BVH*, point3, point3, colour3&);

extern "C" Object* __wrap__ZN3BVH11findNearestEN3glm5tvec3IfLNS0_9precisionE0EEES3_(BVH* self, point3 e, point3 d) {
	Object* h = __real__ZN3BVH11findNearestEN3glm5tvec3IfLNS0_9precisionE0EEES3_(self, e, d);
	g_nNear++;
	bool primary = g_primaryPending;
	g_primaryPending = false;
	if (!primary) {
		if (g_strideNear && (g_nSecondary % g_strideNear) == 0) {
			NearRec r;
			for (int i = 0; i < 3; i++) { r.e[i] = e[i]; r.d[i] = d[i]; }
			r.id = h ? g_ids[h] : -1;
			// re-evaluating the winner's own rayhit with the same ray rewrites its cached hit with the same values
			r.t = h ? h->rayhit(e, d, false) : 0.0f;
			g_near.push_back(r);
		}
		g_nSecondary++;
	}
	return h;
}

extern "C" bool __wrap__ZN3BVH10calcShadowEN3glm5tvec3IfLNS0_9precisionE0EEES3_RS3_(BVH* self, point3 p, point3 l, colour3& shadow) {
	colour3 in = shadow;
	bool lit = __real__ZN3BVH10calcShadowEN3glm5tvec3IfLNS0_9precisionE0EEES3_RS3_(self, p, l, shadow);
	if (g_strideShad && (g_nShad % g_strideShad) == 0 && in[0] == 1.0f && in[1] == 1.0f && in[2] == 1.0f) {
		ShadRec r;
		for (int i = 0; i < 3; i++) { r.p[i] = p[i]; r.l[i] = l[i]; r.shadow[i] = shadow[i]; }
		r.lit = lit ? 1 : 0;
		g_shad.push_back(r);
	}
	g_nShad++;
	return lit;
}

// ---- camera: restatement of q1.cpp:30-53 with the same glm operations --------------------------------
static int vp_width, vp_height;
static point3 eye, facing, camera_right, camera_up;
static void setFacing(float rotationX, float rotationY) {
	const double PI = 3.14159265358979323846264338327950288;
	facing = point3(-sin(rotationY) * cos(rotationX), sin(rotationX), -cos(rotationY) * cos(rotationX));
	for (int i = 0; i < 3; i++)
		if (std::fabs(facing[i]) < 1e-5) facing[i] = 0;
	float aspect_ratio = (float)vp_width / vp_height;
	float h = 1.0f * (float)tan((PI * fov) / 180.0 / 2.0);
	float w = h * aspect_ratio;
	camera_right = glm::normalize(glm::cross(point3(-sin(rotationY), 0, -cos(rotationY)), point3(0, 1, 0))) * w;
	camera_up = glm::normalize(glm::cross(camera_right, facing)) * h;
}
// generalised sub-pixel grid: n=1 gives s(x,y) (q1.cpp:51-53), n=2 gives s_aa (q1.cpp:55-64)
static point3 s_grid(int x, int y, int n, int k) {
	float ox = ((k % n) + 0.5f) / n, oy = ((k / n) + 0.5f) / n;
	return eye + facing + camera_right * (2 * ((x + ox) / vp_width - 0.5f)) + camera_up * (2 * ((y + oy) / vp_height - 0.5f));
}

static void put(FILE* f, const void* p, size_t n) { if (fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(2); } }

int main(int argc, char** argv) {
	if (argc < 5) {
		fprintf(stderr, "usage: ref_driver <scene> <W> <H> <out.bin> [capture_cap=0] [aa_grid=1] [eye x y z rotX rotY]\n");
		return 1;
	}
	const char* sceneName = argv[1];
	vp_width = atoi(argv[2]); vp_height = atoi(argv[3]);
	const char* outPath = argv[4];
	uint64_t cap = argc > 5 ? strtoull(argv[5], 0, 10) : 0;
	int aa = argc > 6 ? atoi(argv[6]) : 1;
	float rotX = 0, rotY = 0;
	eye = point3(0, 0, 0);
	if (argc > 11) { eye = point3(atof(argv[7]), atof(argv[8]), atof(argv[9])); rotX = atof(argv[10]); rotY = atof(argv[11]); }

	// rays mode: ref_driver <scene> 0 0 rays <in.bin> <out.bin> -- BVH::findNearest (bvh.cpp:72) and BVH::calcShadow (bvh.cpp:123) for
	// caller-supplied queries.  in: uint64 n, n x {e[3], d[3]}, uint64 m, m x {p[3], l[3]} (floats); out: n x {int32 id, float t},
	// m x {int32 lit, float shadow[3]} (shadow starts at (1, 1, 1) as in the light classes, objects.cpp:402).
	if (strcmp(outPath, "rays") == 0) {
		if (argc < 7) { fprintf(stderr, "usage: ref_driver <scene> 0 0 rays <in.bin> <out.bin>\n"); return 1; }
		choose_scene(sceneName);
		build_ids();
		FILE* in = fopen(argv[5], "rb");
		FILE* out = fopen(argv[6], "wb");
		if (!in || !out) { perror("rays files"); return 2; }
		uint64_t n = 0, m = 0;
		if (fread(&n, 8, 1, in) != 1) return 2;
		std::vector<float> q(n * 6);
		if (n && fread(q.data(), 24, n, in) != n) return 2;
		for (uint64_t i = 0; i < n; i++) {
			point3 e(q[6 * i], q[6 * i + 1], q[6 * i + 2]), d(q[6 * i + 3], q[6 * i + 4], q[6 * i + 5]);
			Object* h = bvh->findNearest(e, d);
			int32_t hid = h ? g_ids[h] : -1;
			float t = h ? h->rayhit(e, d, false) : 0.0f;
			put(out, &hid, 4); put(out, &t, 4);
		}
		if (fread(&m, 8, 1, in) != 1) return 2;
		q.resize(m * 6);
		if (m && fread(q.data(), 24, m, in) != m) return 2;
		for (uint64_t i = 0; i < m; i++) {
			point3 pt(q[6 * i], q[6 * i + 1], q[6 * i + 2]), l(q[6 * i + 3], q[6 * i + 4], q[6 * i + 5]);
			colour3 shadow(1.0, 1.0, 1.0);
			int32_t lit = bvh->calcShadow(pt, l, shadow) ? 1 : 0;
			float sh[3] = { shadow[0], shadow[1], shadow[2] };
			put(out, &lit, 4); put(out, sh, 12);
		}
		fclose(in); fclose(out);
		return 0;
	}

	// pick mode: ref_driver <scene> <W> <H> pick x y [x y ...] -- the reference's own narration of trace(..., pick = true)
	// (raytracer.cpp:307-327, objects.cpp:54-140) for those pixels, on stdout, one "== pixel x y" header each
	if (strcmp(outPath, "pick") == 0) {
		eye = point3(0, 0, 0);   // (the optional camera arguments above are pixel coordinates in this mode)
		choose_scene(sceneName);
		setFacing(0, 0);
		for (int a = 5; a + 1 < argc; a += 2) {
			int x = atoi(argv[a]), y = atoi(argv[a + 1]);
			std::cout << "== pixel " << x << " " << y << std::endl;
			colour3 colour(0, 0, 0);
			bool h = trace(eye, s_grid(x, y, 1, 0), colour, true);
			std::cout << "== " << (h ? "hit" : "miss") << std::endl;
		}
		return 0;
	}

	choose_scene(sceneName);
	build_ids();
	setFacing(rotX, rotY);

	const size_t npx = (size_t)vp_width * vp_height;
	std::vector<uint8_t> hit(npx);
	std::vector<int32_t> id(npx);
	std::vector<float> tt(npx), rgb(npx * 3), sxyz(npx * 3);
	std::vector<uint32_t> cntNear(npx), cntShad(npx);

	for (int pass = 0; pass < (cap ? 2 : 1); pass++) {
		bool capturing = cap && pass == 1;
		if (capturing) {
			g_strideNear = g_nSecondary / cap + 1;
			g_strideShad = g_nShad / cap + 1;
			srand(1);   // same rand() stream as a fresh process (the reference never seeds)
		}
		g_nNear = g_nShad = g_nSecondary = 0;
		for (int y = 0; y < vp_height; y++) for (int x = 0; x < vp_width; x++) {
			size_t i = (size_t)y * vp_width + x;
			colour3 total(0.0, 0.0, 0.0);
			uint64_t n0 = g_nNear, s0 = g_nShad;
			for (int k = 0; k < aa * aa; k++) {
				point3 s = s_grid(x, y, aa, k);
				if (k == 0) {
					// primary hit record (first sample): straight from BVH::findNearest, then t from the winner's rayhit
					point3 d = s - eye;
					g_primaryPending = true;
					Object* h = bvh->findNearest(eye, d);
					g_nNear--;   // not part of the frame's ray count
					hit[i] = h != NULL;
					id[i] = h ? g_ids[h] : -1;
					tt[i] = h ? h->rayhit(eye, d, false) : 0.0f;
					for (int c = 0; c < 3; c++) sxyz[i * 3 + c] = s[c];
				}
				colour3 colour;
				g_primaryPending = true;
				if (!trace(eye, s, colour, false)) colour = background_colour;   // q1.cpp:135-137,143-145
				total += colour;
			}
			if (aa > 1) total = total / float(aa * aa);   // q1.cpp:140
			for (int c = 0; c < 3; c++) rgb[i * 3 + c] = total[c];
			cntNear[i] = (uint32_t)(g_nNear - n0);
			cntShad[i] = (uint32_t)(g_nShad - s0);
		}
	}

	FILE* f = fopen(outPath, "wb");
	if (!f) { perror(outPath); return 2; }
	char magic[8] = "RTREF02";
	put(f, magic, 8);
	int32_t hdr[8] = { vp_width, vp_height, aa, (int32_t)g_near.size(), (int32_t)g_shad.size(), (int32_t)g_ids.size(), 0, 0 };
	put(f, hdr, sizeof hdr);
	uint64_t counts[2] = { g_nNear, g_nShad };
	put(f, counts, sizeof counts);
	double fovd = fov; put(f, &fovd, 8);
	float cam[15] = { eye[0], eye[1], eye[2], facing[0], facing[1], facing[2], camera_right[0], camera_right[1], camera_right[2],
		camera_up[0], camera_up[1], camera_up[2], background_colour[0], background_colour[1], background_colour[2] };
	put(f, cam, sizeof cam);
	put(f, hit.data(), npx);
	put(f, id.data(), npx * 4);
	put(f, tt.data(), npx * 4);
	put(f, rgb.data(), npx * 12);
	put(f, sxyz.data(), npx * 12);
	put(f, cntNear.data(), npx * 4);
	put(f, cntShad.data(), npx * 4);
	put(f, g_near.data(), g_near.size() * sizeof(NearRec));
	put(f, g_shad.data(), g_shad.size() * sizeof(ShadRec));
	fclose(f);
	fprintf(stderr, "ref_driver: %s %dx%d aa=%d near=%llu shadow=%llu captured %zu/%zu\n", sceneName, vp_width, vp_height, aa,
		(unsigned long long)g_nNear, (unsigned long long)g_nShad, g_near.size(), g_shad.size());
	return 0;
}
