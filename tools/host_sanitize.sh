#!/bin/bash
# The multi-threaded HOST code (scene loader, std_sort_par.h, treelet cuts) under ThreadSanitizer and Address/UB-Sanitizer.  Host only:
# the sanitizers go to the host half of the hipcc line (-Xarch_host), no kernel is launched (rt_load_scene_json + rt_debug_treelet_digest).
# usage: tools/host_sanitize.sh [grid of the height field, default 800 = 1 280 000 triangles: fat leaves with sub-trees]
set -e
GRID=${1:-800}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
cat > $W/main.cpp <<'EOM'
#include "rt_capi.h"
#include <cstdio>
#include <cstdint>
extern "C" int rt_debug_treelet_digest(const rt_flat_scene* s, int threads, uint64_t out[4]);
int main(int argc, char** argv) {
	for (int i = 1; i < argc; i++) {
		rt_flat_scene* s = nullptr;
		if (rt_load_scene_json(argv[i], nullptr, &s)) { printf("load failed: %s\n", rt_last_error()); return 1; }
		uint64_t a[4], b[4];
		const int r1 = rt_debug_treelet_digest(s, 1, a), r2 = rt_debug_treelet_digest(s, 8, b);
		const bool same = a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
		printf("%s: %d triangles, cuts rc %d %d, parallel == sequential %d, treelets %llu + %llu\n", argv[i], s->n_tris, r1, r2, (int)same, (unsigned long long)a[2], (unsigned long long)a[3]);
		rt_flat_scene_free(s);
		if (r1 || r2 || !same) return 1;
	}
	return 0;
}
EOM
python3 - "$ROOT" "$W" "$GRID" <<'EOP'
import sys
sys.path.insert(0, sys.argv[1])
from raytracer_amd import scenes
scenes.c5_scene(sys.argv[2] + "/c5", int(sys.argv[3]), binary=True)
scenes.c2_scene(sys.argv[2] + "/c2", grid=64)
EOP
H=$ROOT/raytracer_amd/csrc
for san in thread address,undefined; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O1 -g -ffp-contract=off -Xarch_host -fsanitize=$san -I$ROOT/include -I$H -Wno-unused-result -pthread $W/main.cpp \
    $H/rt_capi_device.hip $H/rt_kernels.hip $H/host/scene_loader.cpp $H/host/host_api.cpp $H/host/bmp_io.cpp $H/host/raytracer_api.cpp $H/host/view.cpp -o $W/check_${san%%,*}
  echo "== -fsanitize=$san"
  $W/check_${san%%,*} $W/c5/*.json $W/c2/*.json
done
rm -rf $W
echo "host sanitizer runs clean"
