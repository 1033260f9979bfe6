"""raytracer_amd -- MI355X-native trace path behind the reference's Raytracer API.

Python here is a thin mirror of the reference's host interface (raytracer.h:12-18) over the C-ABI in
include/rt_capi.h; the work is done by hand-written HIP kernels (csrc/rt_kernels.hip).

    import raytracer_amd as rt
    rt.choose_scene("scenes/c.json")        # raytracer.cpp:117  (also uploads to the GPU)
    hit, colour = rt.trace(e, s)            # raytracer.cpp:307
    lit, shadow = rt.shadowRay(p, light)    # raytracer.cpp:111
    frame = rt.current_scene().render(640, 480)   # display()'s pixel loop, q1.cpp:130-147
"""
from .api import (FlatScene, Scene, Camera, View, choose_scene, current_scene, trace, shadowRay, fov, background_colour,  # noqa: F401
                  write_bmp, quantize_rgb8, device_count)
from .capi import RtError, MAX_REFLECTIONS, RT_FLAG_COUNTERS, RT_FLAG_TIME_KERNELS, RT_FLAG_WHOLE_FRAME  # noqa: F401
